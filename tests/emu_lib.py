"""Builds and binds the CPU wave-emulator build of the product kernel sources (tests/emu/)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu')
CSRC = os.path.join(ROOT, 'assistive_gym_amd', 'csrc')
_LIBS = {}
# kernel variants (limits + task layer), the same -D sets as csrc/agx_kernels.hip
VARIANT_DEFS = {0: [], 1: ['-DAGX_MAX_DOF=20', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=10', '-DAGX_TASK=1'],
                2: ['-DAGX_MAX_DOF=24', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=12', '-DAGX_ARENA_WORDS=4096', '-DAGX_TASK=2']}


VARIANT_DEFS[3] = ['-DAGX_MAX_DOF=20', '-DAGX_MAX_FREE=1', '-DAGX_MAX_BLOCK=10', '-DAGX_TASK=3']      # dressing (rigid scene; the cloth kernel is a workgroup kernel)
VARIANT_DEFS[4] = ['-DAGX_MAX_DOF=20', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=10', '-DAGX_TASK=4']      # arm manipulation
VARIANT_DEFS['drinking'] = ['-DAGX_MAX_FREE=1', '-DAGX_TASK=5']      # the feeding limits, the drinking task layer, the water kernel (csrc/agx_water.h)
VARIANT_DEFS['drinking_l'] = ['-DAGX_MAX_FREE=1', '-DAGX_MAX_BLOCK=12', '-DAGX_ARENA_WORDS=4040', '-DAGX_TASK=5']      # DrinkingPR2
VARIANT_DEFS['drinking_m'] = ['-DAGX_MAX_FREE=1', '-DAGX_MAX_DOF=20', '-DAGX_MAX_BLOCK=16', '-DAGX_ARENA_WORDS=4040', '-DAGX_TASK=5']      # DrinkingStretch
VARIANT_DEFS['feeding_abs_travel'] = ['-DAGX_NO_REL_TRAVEL']      # narrowphase limits from the per-collider (absolute) travel distances only
VARIANT_DEFS['feeding_trace'] = ['-DAGX_EMU_TRACE_GJK']      # tests/diag/narrowphase_passes.py
VARIANT_DEFS['feeding_trace_sched'] = ['-DAGX_EMU_TRACE_SCHED', '-DAGX_PGS_LV=3']      # tests/diag/solve_schedule_study.py
VARIANT_DEFS['feeding_scan4'] = ['-DAGX_GJK_SCAN_WIDE=0', '-DAGX_GJK_SCAN_ONE=0']      # the 4-per-round support scan of rounds 3-5 (csrc/agx_gjk.h)
VARIANT_DEFS['feeding_reg'] = ['-DAGX_PGS_LV=0']       # the register sweep of csrc/agx_pgs.h (its C++ twin) for the scenes that take the row-local sweep (csrc/agx_pgs_lv.h) by default
VARIANT_DEFS['feeding_lv_cap'] = ['-DAGX_PGS_LV=2', '-DAGX_LV_WINDOW_CAP=300']       # the row-local sweep with a small LDS window: its rows-beyond-the-window path on ordinary scenes
VARIANT_DEFS['feeding_lv2'] = ['-DAGX_PGS_LV=2']       # the row-local sweep with row headers, impulses and velocity slots in LDS (csrc/agx_pgs_lv.h); the default (0) is csrc/agx_pgs_lvw.h
VARIANT_DEFS['feeding_lvs'] = ['-DAGX_PGS_LV=3']       # the row-local sweep with scalar row headers (csrc/agx_pgs_lvs.h), one row per visit: the default of round 5, now the fallback of ...
VARIANT_DEFS['feeding_lvs_cap'] = ['-DAGX_PGS_LV=3', '-DAGX_LV_WINDOW_CAP=300']       # ... with a small LDS window: most rows read their pairs from the scratch record
VARIANT_DEFS['feeding_lvw_cap'] = ['-DAGX_LV_WINDOW_CAP=300']       # the wide row-local sweep (csrc/agx_pgs_lvw.h, the default: variant 0) with a small LDS window
VARIANT_DEFS['feeding_lvw_8steps'] = ['-DAGX_LVW_MAX_STEPS=8']       # ... whose scheduler gives up beyond 8 steps per part: every ordinary substep falls back to the narrow sweep
VARIANT_DEFS['feeding_l'] = ['-DAGX_MAX_COLL=320', '-DAGX_MAX_BLOCK=12', '-DAGX_ARENA_WORDS=4040']
VARIANT_DEFS['feeding_m'] = ['-DAGX_MAX_DOF=20', '-DAGX_MAX_BLOCK=16', '-DAGX_MAX_COLL=320', '-DAGX_ST_WORDS=344', '-DAGX_ARENA_WORDS=4040']    # FeedingStretch
VARIANT_DEFS['bed_m'] = ['-DAGX_MAX_DOF=28', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=16', '-DAGX_ARENA_WORDS=5632', '-DAGX_TASK=1']          # BedBathingStretch
VARIANT_DEFS['scratch_m'] = ['-DAGX_MAX_DOF=28', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=16', '-DAGX_ARENA_WORDS=5632', '-DAGX_TASK=2']      # ScratchItchStretch
VARIANT_DEFS['dressing_m'] = ['-DAGX_MAX_DOF=28', '-DAGX_MAX_FREE=1', '-DAGX_MAX_BLOCK=16', '-DAGX_ARENA_WORDS=5632', '-DAGX_TASK=3']     # DressingStretch (rigid scene)
VARIANT_DEFS['dressing_l'] = ['-DAGX_MAX_DOF=24', '-DAGX_MAX_FREE=1', '-DAGX_MAX_BLOCK=12', '-DAGX_ARENA_WORDS=4096', '-DAGX_TASK=3']
VARIANT_DEFS['arm_l'] = ['-DAGX_MAX_DOF=32', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=22', '-DAGX_ARENA_WORDS=7552', '-DAGX_TASK=4']
VARIANT_DEFS['bed_l'] = ['-DAGX_MAX_DOF=24', '-DAGX_MAX_FREE=2', '-DAGX_MAX_BLOCK=12', '-DAGX_ARENA_WORDS=4096', '-DAGX_TASK=1']
VARIANT_DEFS['settle'] = ['-DAGX_MAX_DOF=48', '-DAGX_MAX_FREE=1', '-DAGX_MAX_BLOCK=48', '-DAGX_ARENA_WORDS=11968', '-DAGX_SCR_ENT=16384', '-DAGX_TASK=1']


def lib(task_kind=0):
    if task_kind not in _LIBS:
        so = os.path.join(EMU, 'libagx_emu_%s.so' % task_kind)
        deps = [os.path.join(EMU, f) for f in ('emu_main.cpp', 'agx_wave.h')] + \
               [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')] + [os.path.join(ROOT, 'include', 'agx_blob.h')]
        def stale():
            return not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps)
        if stale():
            import fcntl
            with open(so + '.lock', 'w') as lock:          # pytest-xdist workers / campaign processes build the same variant at the same time
                fcntl.flock(lock, fcntl.LOCK_EX)
                if stale():
                    tmp = '%s.%d.tmp' % (so, os.getpid())
                    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-I' + EMU, '-I' + CSRC] + VARIANT_DEFS[task_kind] +
                                          ['-o', tmp, os.path.join(EMU, 'emu_main.cpp')])
                    os.replace(tmp, so)
        L = C.CDLL(so)
        L.agx_emu_run.restype = C.c_int
        _LIBS[task_kind] = L
    return _LIBS[task_kind]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Emu:
    def __init__(self, blob, kind=None):
        self.blob = blob
        self.L = lib(kind) if kind is not None else lib('drinking_m' if blob.ndof > 16 else 'drinking_l' if blob.nrobot > 10 else 'drinking') if blob.task_kind == 5 else lib('settle' if blob.ndof > 32 else 'arm_l' if (blob.task_kind == 4 and blob.ndof > 20) else 'feeding_m' if (blob.task_kind == 0 and blob.ndof > 16) else 'feeding_l' if (blob.task_kind == 0 and blob.h['NCOLL'] > 256) else ('bed_m' if blob.task_kind == 1 and blob.nrobot > 12 else 'scratch_m' if blob.task_kind == 2 and blob.nrobot > 12 else 'bed_l' if blob.task_kind == 1 and (blob.ndof > 20 or blob.nrobot > 10) else 'dressing_m' if blob.task_kind == 3 and blob.nrobot > 12 else 'dressing_l' if blob.task_kind == 3 and (blob.ndof > 20 or blob.nrobot > 10) else blob.task_kind))
        self.words = np.ascontiguousarray(blob.words)
        lay = (C.c_int * 8)()
        self.L.agx_emu_debug_layout(lay)
        self.DEBUG_WORDS, self.DBG_CON, self.DBG_MINV, self.MINV_STRIDE, self.DBG_HDR, self.DBG_LAM, self.DBG_TIME, self.DBG_QDD = list(lay)

    def _run(self, state, action, mode, nsettle, debug=False):
        obs = np.zeros(self.blob.obs_dim, dtype=np.float32)
        rew = np.zeros(1, dtype=np.float32)
        done = np.zeros(4, dtype=np.uint8)
        info = np.zeros(8, dtype=np.float32)
        dbg = np.zeros(self.DEBUG_WORDS, dtype=np.float32) if debug else None
        act = np.ascontiguousarray(action if action is not None else np.zeros(self.blob.act_dim), dtype=np.float32)
        rc = self.L.agx_emu_run(_p(self.words), _p(state), _p(act), _p(obs), _p(rew), _p(done), _p(info), _p(dbg), C.c_int(mode), C.c_int(nsettle))
        assert rc == 0, 'wave emulator reported divergent control flow'
        return obs, float(rew[0]), bool(done[0]), info, dbg

    def manifold_get(self):
        out = np.zeros((64, 12))
        self.L.agx_emu_manifold_get.restype = C.c_int
        return out[:self.L.agx_emu_manifold_get(_p(out), C.c_int(64))].copy()

    def manifold_set(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        self.L.agx_emu_manifold_set(_p(rows), C.c_int(len(rows)))

    def forget_warm(self):
        """the warm-start memory (AGX_P_WARMSTART) of the emulated environment's scratch record: cleared, as agx_set_state / the resets do"""
        self.L.agx_emu_forget_warm()

    def step(self, state, action, debug=False):
        return self._run(state, action, 0, 0, debug)

    def step_water(self, state, water, action):
        """one env step of the drinking scene (state and water in place) -> (obs, reward, done, info)"""
        assert state.dtype == np.float32 and water.dtype == np.float32 and water.flags.c_contiguous
        obs = np.zeros(self.blob.obs_dim, dtype=np.float32); rew = np.zeros(1, dtype=np.float32); done = np.zeros(1, dtype=np.uint8); info = np.zeros(8, dtype=np.float32)
        rc = self.L.agx_emu_step_water(_p(self.words), _p(state), _p(water), _p(np.ascontiguousarray(action, dtype=np.float32)), _p(obs), _p(rew), _p(done), _p(info))
        assert rc == 0, 'wave emulator reported divergent control flow'
        return obs, float(rew[0]), bool(done[0]), info

    def settle(self, state, n, debug=False):
        return self._run(state, None, 1, n, debug)[4]

    def sample(self, seed, impairment_mode=-1, gender_mode=-1, settled=None, fell=None):
        """device-side reset generator (csrc/agx_reset.h) for one env -> (state record, info[4]); settled: the rag-doll model's settled
        record of this environment (bed bathing)"""
        st = np.zeros(self.blob.state_words, dtype=np.float32)
        info = np.zeros(4, dtype=np.float32)
        settled = None if settled is None else np.ascontiguousarray(settled, dtype=np.float32)
        fell = None if fell is None else np.ascontiguousarray(fell, dtype=np.float32)
        rc = self.L.agx_emu_sample(_p(self.words), _p(st), C.c_uint64(seed), C.c_int(impairment_mode), C.c_int(gender_mode), _p(info), _p(settled), _p(fell))
        assert rc == 0, 'wave emulator reported divergent control flow'
        return st, info

    def check_collisions(self, state):
        """agx_check_collisions for one env: AGX_COLLIDE_* flags (the state is not advanced)"""
        st = np.ascontiguousarray(state, dtype=np.float32).copy()
        f = self.L.agx_emu_check_collisions(_p(self.words), _p(st))
        assert f >= 0, 'wave emulator reported divergent control flow'
        assert np.array_equal(st.view(np.uint32), np.ascontiguousarray(state, dtype=np.float32).view(np.uint32)), 'the collision pass must not change the state'
        return f

    def observe(self, state):
        return self._run(state, None, 2, 0)[0]
