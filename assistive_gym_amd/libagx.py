"""ctypes binding of libagx (include/agx.h).  There is no CPU fallback: if the HIP library is
missing or no GPU is visible, constructing a stepper raises."""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  -- imported BEFORE libagx.so is opened: torch ships its own libamdhip64 and the
#                process must end up with ONE HIP runtime (libagx binds to the already loaded one by soname)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AGX_LIB', os.path.join(HERE, 'lib', 'libagx.so'))   # AGX_LIB: A/B builds of the same library
_LIB = None

EXPORTS = ['agx_version', 'agx_last_error', 'agx_device_count', 'agx_lds_bytes_per_env', 'agx_create', 'agx_destroy', 'agx_dims',
           'agx_set_state', 'agx_get_state', 'agx_state_dev', 'agx_settle', 'agx_settle_debug', 'agx_cloth_nodes', 'agx_set_cloth', 'agx_get_cloth', 'agx_cloth_dev', 'agx_set_cloth_pool', 'agx_get_cloth_report', 'agx_step', 'agx_step_debug', 'agx_step_timed', 'agx_debug_words',
           'agx_observe', 'agx_observe_masked', 'agx_reset_done_at', 'agx_sample_reset', 'agx_reset', 'agx_attach_settle_model', 'agx_reset_done', 'agx_step_host', 'agx_observe_host', 'agx_profile_begin', 'agx_profile_end',
           'agx_synchronize', 'agx_selftest', 'agx_debug_layout', 'agx_variant_name', 'agx_overflow_count', 'agx_set_env_offset', 'agx_check_collisions',
           'agx_comm_unique_id', 'agx_comm_init_rank', 'agx_comm_destroy', 'agx_allgather', 'agx_pack_step']


class AgxError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AgxError('libagx.so is not built (%s); run `python -m assistive_gym_amd.build` -- there is no CPU path' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.agx_version.restype = C.c_char_p
        L.agx_last_error.restype = C.c_char_p
        L.agx_variant_name.restype = C.c_char_p
        _LIB = L
    return _LIB


def check(rc, what=''):
    if rc != 0:
        raise AgxError('%s failed (%d): %s' % (what, rc, load().agx_last_error().decode()))


def comm_unique_id():
    """agx_comm_unique_id: the 128-byte RCCL id rank 0 obtains and hands to the other ranks (by the host's own means)"""
    buf = C.create_string_buffer(128)
    check(load().agx_comm_unique_id(buf), 'agx_comm_unique_id')
    return bytes(buf.raw)


def comm_init_rank(device, rank, world, unique_id):
    """agx_comm_init_rank (collective over all ranks) -> communicator handle for Stepper.allgather"""
    assert len(unique_id) == 128
    comm = C.c_void_p()
    check(load().agx_comm_init_rank(C.c_int(device), C.c_int(rank), C.c_int(world), C.c_char_p(unique_id), C.byref(comm)), 'agx_comm_init_rank')
    return comm.value


def comm_destroy(comm):
    if comm:
        check(load().agx_comm_destroy(C.c_void_p(comm)), 'agx_comm_destroy')


def _ptr(x):
    """device pointer of a torch tensor / raw int, or None"""
    if x is None:
        return None
    if hasattr(x, 'data_ptr'):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Stepper:
    """Thin owner of one agx_handle: N lock-stepped environments on one GPU."""

    def __init__(self, blob, n_envs, device=0):
        self.L = load()
        self.blob = blob
        words = np.ascontiguousarray(blob.words)
        h = C.c_void_p()
        check(self.L.agx_create(words.ctypes.data_as(C.c_void_p), C.c_size_t(words.nbytes), C.c_int(n_envs), C.c_int(device), C.byref(h)), 'agx_create')
        self.h = h
        self.n_envs, self.device = n_envs, device
        self.act_dim, self.obs_dim, self.state_words = blob.act_dim, blob.obs_dim, blob.state_words

    def close(self):
        if getattr(self, 'h', None):
            self.L.agx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_chunks(self):
        """independent chunks of environments a step is issued as (AGX_CHUNKS, default 3 from 2048 environments on)"""
        e = os.environ.get('AGX_CHUNKS')
        from .model import compiler as L
        particles = self.blob.h.get('OFF_CLOTH', 0) and int(self.blob.i[self.blob.h['OFF_CLOTH'] + L.CL['PARTICLES']])     # the drinking scenes run unchunked (agx_api.hip)
        nc = int(e) if e else (1 if particles else (3 if self.n_envs >= 2048 else 1))
        nc = min(max(nc, 1), 8)
        return 1 if self.n_envs < 64 * nc else nc

    def debug_layout(self):
        """[words per env, contacts offset, M^-1 offset, M^-1 row stride, row headers, impulses, phase timers, qdd] of the debug record"""
        out = (C.c_int * 8)()
        check(self.L.agx_debug_layout(self.h, out), 'agx_debug_layout')
        return list(out)

    def variant(self):
        return self.L.agx_variant_name(self.h).decode()

    def overflow_count(self):
        """contacts dropped by a budget since the stepper was created (0 in a healthy run)"""
        out = C.c_int()
        check(self.L.agx_overflow_count(self.h, C.byref(out)), 'agx_overflow_count')
        return out.value

    def check_collisions(self):
        """AGX_COLLIDE_* flags of every environment's current state (uint8 [n_envs]); the states are not advanced"""
        out = np.zeros(self.n_envs, dtype=np.uint8)
        check(self.L.agx_check_collisions(self.h, out.ctypes.data_as(C.c_void_p), None), 'agx_check_collisions')
        return out

    def set_env_offset(self, env_offset):
        """global index of this stepper's first env (multi-GPU sharding): keeps agx_reset_done's pool draw placement independent"""
        check(self.L.agx_set_env_offset(self.h, C.c_longlong(int(env_offset))), 'agx_set_env_offset')

    def pack_step(self, obs, reward, done, info, packed, stream=0):
        """agx_pack_step: [n_envs, obs_dim + 4] = observation | reward | done | total_force_on_human | task_success (float32 device tensors; done uint8)"""
        assert packed.shape == (self.n_envs, obs.shape[1] + 4) and packed.is_contiguous() and obs.is_contiguous()
        check(self.L.agx_pack_step(self.h, _ptr(obs), _ptr(reward), _ptr(done), _ptr(info), _ptr(packed), C.c_void_p(stream)), 'agx_pack_step')

    def allgather(self, local, gathered, comm=None, stream=0):
        """agx_allgather of a float32 device tensor (comm None = single rank)"""
        check(self.L.agx_allgather(self.h, _ptr(local), _ptr(gathered), C.c_size_t(local.numel()), C.c_void_p(comm), C.c_void_p(stream)), 'agx_allgather')

    def set_state(self, states):
        states = np.ascontiguousarray(states, dtype=np.float32)
        assert states.shape == (self.n_envs, self.state_words)
        check(self.L.agx_set_state(self.h, states.ctypes.data_as(C.c_void_p)), 'agx_set_state')

    def get_state(self):
        out = np.zeros((self.n_envs, self.state_words), dtype=np.float32)
        check(self.L.agx_get_state(self.h, out.ctypes.data_as(C.c_void_p)), 'agx_get_state')
        return out

    # ---- models with a cloth section: the garments, float32 [n_envs, 2, nodes, 3] (positions, velocities)
    def cloth_nodes(self):
        n = C.c_int()
        check(self.L.agx_cloth_nodes(self.h, C.byref(n)), 'agx_cloth_nodes')
        return n.value

    def set_cloth(self, cloth):
        cloth = np.ascontiguousarray(cloth, dtype=np.float32)
        assert cloth.shape == (self.n_envs, 2, self.cloth_nodes(), 3)
        check(self.L.agx_set_cloth(self.h, cloth.ctypes.data_as(C.c_void_p)), 'agx_set_cloth')

    def get_cloth(self):
        out = np.zeros((self.n_envs, 2, self.cloth_nodes(), 3), dtype=np.float32)
        check(self.L.agx_get_cloth(self.h, out.ctypes.data_as(C.c_void_p)), 'agx_get_cloth')
        return out

    def get_cloth_report(self):
        """the cloth kernel's report of the last step, float32 [n_envs, 20 + 2 * slots * nodes]: sleeve vertices (18), 2 unused, then per node and
        contact slot {node height, |contact force| of the last substep or -1} (agx_get_cloth_report)"""
        w = C.c_int()
        check(self.L.agx_get_cloth_report(self.h, None, C.byref(w)), 'agx_get_cloth_report')
        out = np.zeros((self.n_envs, w.value), dtype=np.float32)
        check(self.L.agx_get_cloth_report(self.h, out.ctypes.data_as(C.c_void_p), None), 'agx_get_cloth_report')
        return out

    def set_cloth_pool(self, pool_cloth):
        """pool_cloth: float32 device tensor [pool_n, 2, nodes, 3]; the caller keeps it alive"""
        check(self.L.agx_set_cloth_pool(self.h, _ptr(pool_cloth)), 'agx_set_cloth_pool')

    def state_dev(self):
        p = C.c_void_p()
        check(self.L.agx_state_dev(self.h, C.byref(p)), 'agx_state_dev')
        return p.value

    def state_tensor(self):
        """the state records as a torch tensor over the handle's own device memory (float32 [n_envs, state_words], no copy)"""
        import torch

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = dict(shape=(self.n_envs, self.state_words), typestr='<f4', data=(self.state_dev(), False), version=2, strides=None)
        return torch.as_tensor(v, device='cuda:%d' % self.device)

    def cloth_tensor(self):
        """the garments as a torch tensor over the handle's own device memory (float32 [n_envs, 2, nodes, 3], no copy)"""
        import torch
        p = C.c_void_p()
        check(self.L.agx_cloth_dev(self.h, C.byref(p)), 'agx_cloth_dev')

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = dict(shape=(self.n_envs, 2, self.cloth_nodes(), 3), typestr='<f4', data=(p.value, False), version=2, strides=None)
        return torch.as_tensor(v, device='cuda:%d' % self.device)

    def settle(self, n_substeps, stream=0):
        check(self.L.agx_settle(self.h, C.c_int(n_substeps), C.c_void_p(stream)), 'agx_settle')

    def settle_debug(self, n_substeps, debug, stream=0):
        check(self.L.agx_settle_debug(self.h, C.c_int(n_substeps), _ptr(debug), C.c_void_p(stream)), 'agx_settle_debug')

    def step_dev(self, actions, obs, reward, done, info=None, stream=0, debug=None):
        if debug is not None:
            check(self.L.agx_step_debug(self.h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(done), _ptr(info), _ptr(debug), C.c_void_p(stream)), 'agx_step_debug')
        else:
            check(self.L.agx_step(self.h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(done), _ptr(info), C.c_void_p(stream)), 'agx_step')

    def step_timed(self, actions, obs, reward, done, info=None, stream=0):
        """one step with HIP events after the launches; returns (summed ms, launch counts) of the build, solve, finish kernels"""
        ms, cnt = (C.c_float * 3)(), (C.c_int * 3)()
        check(self.L.agx_step_timed(self.h, _ptr(actions), _ptr(obs), _ptr(reward), _ptr(done), _ptr(info), C.c_void_p(stream), ms, cnt), 'agx_step_timed')
        return [ms[0], ms[1], ms[2]], [cnt[0], cnt[1], cnt[2]]

    def observe_dev(self, obs, stream=0, mask=None):
        """mask: uint8 device tensor, only those environments' rows are written (agx_observe_masked)"""
        check(self.L.agx_observe_masked(self.h, _ptr(obs), _ptr(mask), C.c_void_p(stream)), 'agx_observe_masked')

    IMPAIRMENT_MODES = {'random': -1, 'no_tremor': -2, 'none': 0, 'limits': 1, 'weakness': 2, 'tremor': 3}
    GENDER_MODES = {'random': -1, 'male': 0, 'female': 1}

    def sample_reset(self, seed, impairment='random', gender='random', ik_info=None, stream=0):
        """device-side FeedingEnv.reset sampling of every env (env i from seed + i); follow with settle(25)"""
        check(self.L.agx_sample_reset(self.h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_int(self.IMPAIRMENT_MODES[impairment]),
                                      C.c_int(self.GENDER_MODES[gender]), _ptr(ik_info), C.c_void_p(stream)), 'agx_sample_reset')

    def reset(self, mask=None, seeds=None, seed=0, impairment='random', gender='random', settle_substeps=25, stream=0):
        """reset() of the envs selected by `mask` (uint8 device tensor, None = all): device-side sampling from
        `seeds` (uint64/int64 device tensor) or seed + env index, then the settle substeps on those envs only"""
        check(self.L.agx_reset(self.h, _ptr(mask), _ptr(seeds), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_int(self.IMPAIRMENT_MODES[impairment]),
                               C.c_int(self.GENDER_MODES[gender]), C.c_int(settle_substeps), C.c_void_p(stream)), 'agx_reset')

    def attach_settle_model(self, other, n_substeps):
        """bed bathing: `other` is a Stepper on the rag-doll model (bed_settle) with as many environments; sample_reset / reset of this stepper
        then drop and settle that model's humans first (bed_bathing.py:119-137) and read their resting poses.  Keeps `other` alive."""
        check(self.L.agx_attach_settle_model(self.h, other.h if other is not None else None, C.c_int(n_substeps)), 'agx_attach_settle_model')
        self._settle_model = other

    def reset_done(self, pool, pool_n, done, stream=0, iteration=-1):
        """iteration >= 0: the replacement states join the batch at that iteration (agx_reset_done_at)"""
        check(self.L.agx_reset_done_at(self.h, _ptr(pool), C.c_int(pool_n), _ptr(done), C.c_int(iteration), C.c_void_p(stream)), 'agx_reset_done_at')

    def step_host(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_envs, self.act_dim)
        obs = np.zeros((self.n_envs, self.obs_dim), dtype=np.float32)
        rew = np.zeros(self.n_envs, dtype=np.float32)
        done = np.zeros(self.n_envs, dtype=np.uint8)
        info = np.zeros((self.n_envs, 8), dtype=np.float32)
        check(self.L.agx_step_host(self.h, a.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p), rew.ctypes.data_as(C.c_void_p),
                                   done.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p)), 'agx_step_host')
        return obs, rew, done.astype(bool), info

    def observe_host(self):
        obs = np.zeros((self.n_envs, self.obs_dim), dtype=np.float32)
        check(self.L.agx_observe_host(self.h, obs.ctypes.data_as(C.c_void_p)), 'agx_observe_host')
        return obs

    def profile_begin(self, stream=0):
        check(self.L.agx_profile_begin(self.h, C.c_void_p(stream)), 'agx_profile_begin')

    def profile_end(self, stream=0):
        ms = C.c_float()
        check(self.L.agx_profile_end(self.h, C.c_void_p(stream), C.byref(ms)), 'agx_profile_end')
        return ms.value

    def synchronize(self, stream=0):
        check(self.L.agx_synchronize(self.h, C.c_void_p(stream)), 'agx_synchronize')
