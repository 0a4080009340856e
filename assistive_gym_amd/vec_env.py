"""Batched environments on one GPU (torch tensors in / out, state resident in HBM).

The scalar gym.Env facade with the reference's class names lives in assistive_gym_amd/envs.py;
this is the data-parallel form: N lock-stepped copies of FeedingJacoEnv (assistive_gym/envs/feeding_envs.py:29-31)
or BedBathingSawyerEnv (assistive_gym/envs/bed_bathing_envs.py:23-25), stepped by a few kernel launches per env.step().
"""
import numpy as np
import torch

from .blob import ModelBlob
from .host.reset import make_states
from .libagx import Stepper
from .shard import episode_seed, pool_indices

SETTLE_STEPS = 25   # feeding.py:178-179
DRESSING_SETTLE_STEPS = 50   # dressing.py:186-187
RAGDOLL_SETTLE_STEPS = 100   # bed_bathing.py:130-131
DRINKING_SETTLE_STEPS = 50   # drinking.py:176-177: the water drops into the cup


def attach_ragdoll_model(stepper, n_envs, device):
    """bed bathing: the stepper's reset generator reads the human's resting pose from the settled records of a second model, the rag doll
    (bed_bathing.py:119-137): a Stepper on `bed_settle` with as many environments, attached through agx_attach_settle_model"""
    rag = Stepper(ModelBlob.load('bed_settle'), n_envs, device)
    stepper.attach_settle_model(rag, RAGDOLL_SETTLE_STEPS)
    return rag


ARM_FALL_STEPS = 100         # arm_manipulation.py:145-146
ARM_DROP_BASE = (-0.25, 0.2, 0.95)   # arm_manipulation.py:123


def attach_arm_fall_models(stepper, blob, n_envs, device):
    """arm manipulation: ArmManipulationEnv.reset has TWO settles (arm_manipulation.py:117-146) -- the rag doll (bed_settle,
    dropped from its own spot) and the fall of the posed right arm, which runs on the task's own model at the reset's gravity of -1
    (ModelBlob.fall_model()).  The fall handle is attached to the task's, the rag doll to the fall handle.  -> (fall, ragdoll) steppers"""
    rag = Stepper(ModelBlob.load('bed_settle').with_drop_base(ARM_DROP_BASE), n_envs, device)
    fall = Stepper(blob.fall_model(), n_envs, device)
    fall.attach_settle_model(rag, RAGDOLL_SETTLE_STEPS)
    stepper.attach_settle_model(fall, ARM_FALL_STEPS)
    return fall, rag


def build_reset_pool(blob, pool_size, seed, device=0, impairment='random', sampler='device', _depth=0):
    """pool_size post-reset states: FeedingEnv.reset's sampling (sampler 'device': agx_sample_reset on the GPU;
    'host': the numpy path of host/reset.py), then the 25 settle steps of feeding.py:178-179 on the device.
    BedBathingSawyer: BedBathingEnv.reset restated on the host (host/reset_bed.py) around the rag-doll settle on the device.
    Returns a float32 (pool_size, state_words) array."""
    from .model import compiler as L
    if blob.task_kind == L.TASK_BED_BATHING and blob.has_reset_generator and sampler == 'device':
        # BedBathingEnv.reset on the device: the rag doll's drop record, its 100-step settle, then the sampler of this model (base pose search,
        # tool, targets) reading the resting pose -- agx_sample_reset with the rag-doll model attached
        st = Stepper(blob, pool_size, device)
        rag = attach_ragdoll_model(st, pool_size, device)
        st.sample_reset(seed, impairment=impairment)
        st.synchronize()
        out = st.get_state()
        st.close(); rag.close()
        return out
    if blob.task_kind == L.TASK_BED_BATHING:
        from .host.reset_bed import make_states as make_bed_states, RagdollSettler
        # the rag-doll settle of BedBathingEnv.reset runs on the device (bed_settle kernel variant), the rest on the host
        from .host.reset_bed import DeviceCollisionChecker
        return make_bed_states(blob, pool_size, seed=seed, impairment=impairment, settler=RagdollSettler(pool_size, device),
                               checker=DeviceCollisionChecker(blob, pool_size, device))[0]
    if blob.task_kind == L.TASK_ARM_MANIPULATION and blob.has_reset_generator and sampler == 'device':
        # ArmManipulationEnv.reset on the device: rag doll, the arm's fall in the fall model, then this model's sampler (one base pose for both arms of PR2 / Baxter)
        st = Stepper(blob, pool_size, device)
        fall, rag = attach_arm_fall_models(st, blob, pool_size, device)
        st.sample_reset(seed, impairment='no_tremor' if impairment == 'random' else impairment)
        st.synchronize()
        out = st.get_state()
        st.close(); fall.close(); rag.close()
        return out
    if blob.task_kind == L.TASK_ARM_MANIPULATION:
        # ArmManipulationEnv.reset (host/reset_arm.py) around its two settles on the device: the rag doll, then the fall of the right arm
        from .host.reset_arm import make_states as make_arm_states, ArmFallSettler
        from .host.reset_bed import RagdollSettler, DeviceCollisionChecker
        return make_arm_states(blob, pool_size, seed=seed, impairment='no_tremor' if impairment == 'random' else impairment,
                               settler=RagdollSettler(pool_size, device), arm_settler=ArmFallSettler(blob, pool_size, device),
                               checker=DeviceCollisionChecker(blob, pool_size, device))[0]
    if blob.task_kind == L.TASK_SCRATCH_ITCH and blob.has_reset_generator and sampler == 'device':
        # a wheelchair-mounted arm (Jaco, Panda) or a free-standing PR2 / Baxter (base pose search on the device): ScratchItchEnv.reset sampled by
        # the device-side reset generator, as for the feeding scenes
        st = Stepper(blob, pool_size, device)
        st.sample_reset(seed, impairment=impairment)
        st.synchronize()
        out = st.get_state()
        st.close()
        return out
    if blob.task_kind == L.TASK_SCRATCH_ITCH:
        from .host.reset_scratch import make_states as make_scratch_states
        from .host.reset_bed import DeviceCollisionChecker
        return make_scratch_states(blob, pool_size, seed=seed, impairment=impairment, checker=DeviceCollisionChecker(blob, pool_size, device))[0]
    if blob.task_kind == L.TASK_DRESSING and blob.has_reset_generator and sampler == 'device':
        # DressingEnv.reset on the device: sampling (base pose search for Baxter / PR2), the garment at the end effector, the 50-step settle
        # under half gravity, full gravity afterwards -- all inside agx_reset
        st = Stepper(blob, pool_size, device)
        st.reset(None, None, seed, impairment=impairment, settle_substeps=DRESSING_SETTLE_STEPS)
        st.synchronize()
        out = st.get_state(), st.get_cloth()
        st.close()
        return out
    if blob.task_kind == L.TASK_DRINKING and sampler == 'device':
        # DrinkingEnv.reset on the device (drinking.py:122-181): sampling (IK restarts / base pose search / placement draws), the water grid above
        # the cup, the 50 steps in which it drops into the cup -- all inside agx_reset; returns (states, water): the particles of a pool
        # entry travel with its state record like a garment
        assert blob.has_reset_generator
        st = Stepper(blob, pool_size, device)
        st.reset(None, None, seed, impairment=impairment, settle_substeps=DRINKING_SETTLE_STEPS)
        st.synchronize()
        out = st.get_state(), st.get_cloth()
        st.close()
        return out
    if blob.task_kind == L.TASK_DRINKING:
        # the numpy sampler (host/reset_drinking.py: the wheelchair-mounted arms) around the device's 50-step settle
        from .host.reset_drinking import make_states as make_drinking_states
        states, water, _ = make_drinking_states(blob, pool_size, seed=seed, impairment=impairment)
        st = Stepper(blob, pool_size, device)
        st.set_state(states); st.set_cloth(water)
        st.settle(DRINKING_SETTLE_STEPS)
        st.synchronize()
        out = st.get_state(), st.get_cloth()
        st.close()
        return out
    if blob.task_kind == L.TASK_DRESSING:
        # DressingEnv.reset restated on the host (host/reset_dressing.py) around the 50-step cloth settle on the device;
        # returns (states, garments): the garment of a pool entry travels with its state record
        from .host.reset_dressing import make_states as make_dressing_states, ClothSettler
        from .host.reset_bed import DeviceCollisionChecker
        st, cloth, _ = make_dressing_states(blob, pool_size, seed=seed, impairment=impairment, settler=ClothSettler(blob, pool_size, device),
                                            checker=DeviceCollisionChecker(blob, pool_size, device))
        return st, cloth
    st = Stepper(blob, pool_size, device)
    if blob.meta.get('mount') in ('toc', 'mobile') and not blob.has_reset_generator:      # a mobile robot (Stretch): placed by the numpy sampler (older blobs: base pose search on the host)
        sampler = 'host'
    if sampler == 'device':
        st.sample_reset(seed, impairment=impairment)
    elif sampler == 'host':
        checker = None
        if blob.meta.get('mount') in ('toc', 'mobile'):
            from .host.reset_bed import DeviceCollisionChecker
            checker = DeviceCollisionChecker(blob, pool_size, device)
        states, _ = make_states(blob, pool_size, seed=seed, impairment=impairment, checker=checker)
        st.set_state(states)
    else:
        raise ValueError("sampler must be 'device' or 'host'")
    st.settle(SETTLE_STEPS)
    st.synchronize()
    out = st.get_state()
    st.close()
    # A sampled start whose arm pose the IK restarts left deep inside the table or the bowl is pushed out by the contact rows with whatever
    # velocity closes the gap in one substep (no penetration-recovery clamp as in Bullet's split impulse): a few in a thousand blow up during
    # the settle.  Such entries never enter a pool: they are drawn again from the seeds after this batch's.
    bad = ~np.isfinite(out[:, :blob.h['S_ENV']]).all(axis=1)          # (the float part of a record: the env / task words hold integers)
    if bad.any() and _depth < 4:
        out[bad] = build_reset_pool(blob, int(bad.sum()), seed + pool_size, device, impairment, sampler, _depth + 1)
    elif bad.any():
        raise RuntimeError('%d of %d sampled reset states are not finite after the settle (seed %d)' % (bad.sum(), pool_size, seed))
    return out


def _refresh_worker(model, coop, seed, batch, device, impairment, sampler, queue):
    """child process of PoolRefresher: post-reset states batch after batch, each from its own seeds (reset sampling on the host or the
    device + the settles on the device, exactly as the first pool was built: build_reset_pool)"""
    blob = ModelBlob.load(model)
    if coop and not blob.is_coop:
        blob = blob.coop()
    g = 1
    while True:
        out = build_reset_pool(blob, batch, int(seed) + 1_000_003 * g, device, impairment, sampler)
        queue.put((g, out))               # blocks while two batches are waiting: the worker never runs far ahead of the rollout
        g += 1


class PoolRefresher:
    """Fresh start states for the tasks whose reset is sampled on the host (free-standing robots: base pose search; the rag-doll and
    cloth settles): a child process keeps producing batches of post-reset states, the rollout swaps them into the pool at episode
    boundaries.  The reference draws a new human / target / base pose at EVERY reset (bed_bathing.py:112-171); a fixed pool shows a run
    the same 256 states forever -- this bounds the repetition by the host sampler's rate instead (a few states per second per process;
    the on-device generator of the wheelchair-mounted scenes, reset='device', needs none of this).
    sync=False: batches are taken when they are ready (the pool's content then depends on timing); sync=True: episode e waits for batch
    e, so that a run is reproducible and ranks that shard a batch see the same pools."""

    def __init__(self, model, coop, seed, batch, device, impairment, sampler, sync=False):
        import multiprocessing as mp
        ctx = mp.get_context('spawn')
        self.queue = ctx.Queue(maxsize=2)
        self.sync, self.batch, self.taken = sync, batch, 0
        self.proc = ctx.Process(target=_refresh_worker, args=(model, coop, seed, batch, device, impairment, sampler, self.queue), daemon=True)
        self.proc.start()

    def poll(self):
        """batches that are due: [states or (states, garments)]"""
        import queue as _q
        out = []
        if self.sync:                                               # episode e waits for batch e
            while True:
                try:
                    out.append(self.queue.get(timeout=5)[1])
                    break
                except _q.Empty:
                    if not self.proc.is_alive():
                        raise RuntimeError('the pool refresher process has died (exit code %s)' % self.proc.exitcode)
        else:
            while True:
                try:
                    out.append(self.queue.get_nowait()[1])
                except _q.Empty:
                    break
        self.taken += len(out)
        return out

    def close(self):
        if self.proc.is_alive():
            self.proc.terminate()
        self.proc.join(timeout=5)


class AssistiveVecEnv:
    """N lock-stepped environments of one compiled model (`model` = blob name: 'feeding_jaco', 'bed_bathing_sawyer').
    reset modes (all sampled and settled on the GPU unless 'host'):
      'pool'    -- a fixed pool of pool_size post-reset states generated once; done envs draw from it
                   (BASELINE config 2: "auto-reset from pool");
      'device'  -- every episode of every env starts from a NEWLY sampled state, as in the reference where each
                   reset() redraws the human, the IK start pose, the bowl ... (feeding.py:114-182): agx_reset
                   (sampling + settle, in place) when the lock-stepped batch reaches the end of its 200-step episode;
      'host'    -- 'pool' with the numpy sampler (host/reset.py)."""

    model = 'feeding_jaco'

    coop = False

    def __init__(self, n_envs, device=0, seed=1001, pool_size=256, blob=None, impairment='random', auto_reset=True, reset='pool', model=None, coop=None,
                 pool_refresh=0, pool_refresh_sync=False):
        assert reset in ('pool', 'device', 'host')
        self.blob = blob or ModelBlob.load(model or self.model)
        if (self.coop if coop is None else coop) and not self.blob.is_coop:
            self.blob = self.blob.coop()
        self.n_envs, self.device_index, self.seed = n_envs, device, seed
        self.device = torch.device('cuda', device)
        from .model import compiler as _L
        if self.blob.task_kind == _L.TASK_ARM_MANIPULATION and impairment == 'random':
            impairment = 'no_tremor'                      # build_assistive_env(human_impairment='no_tremor'), arm_manipulation.py:112
        self.pool_size, self.impairment, self.auto_reset, self.reset_mode = pool_size, impairment, auto_reset, reset
        self.stepper = Stepper(self.blob, n_envs, device)
        self.act_dim, self.obs_dim = self.blob.act_dim, self.blob.obs_dim
        self.obs = torch.zeros((n_envs, self.obs_dim), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(n_envs, dtype=torch.float32, device=self.device)
        self.done = torch.zeros(n_envs, dtype=torch.uint8, device=self.device)
        self.info = torch.zeros((n_envs, 8), dtype=torch.float32, device=self.device)
        self.pool = None
        self.episode_len = int(self.blob.task_f('EPISODE_LEN'))
        self.env_offset, self._t, self._episode = 0, 0, 0
        self.terminal_obs = None
        self.keep_terminal_obs = False                # adapters (rllib.py): terminal_obs after EVERY step = the step's observation with the rows that ended
                                                      # holding their LAST observation, self.obs holding the first one of their new episode
        # pool_refresh = k > 0: a child process samples k new start states at a time, swapped into the pool at episode boundaries (PoolRefresher)
        self.pool_refresh, self.pool_refresh_sync, self._refresher, self._refresh_cursor, self.pool_refreshed = int(pool_refresh), pool_refresh_sync, None, 0, 0
        self._model_name = model or self.model
        self.start_states_redrawn = 0                 # reset='device': environments whose sampled start came out of the settle implausible and were drawn again
        # reset='device' of the models whose reset has settles of its own before the sampling: further handles attached to the stepper
        # (bed bathing: the rag doll; arm manipulation: the arm's fall, and the rag doll behind it) -- for whichever class or model name built this env
        self._aux_steppers = ()
        if self.reset_mode == 'device' and self.blob.has_reset_generator:
            from .model import compiler as L
            flags = int(self.blob.i[int(self.blob.i[L.H['OFF_RESET']]) + L.X_['FLAGS']])
            if self.blob.task_kind == L.TASK_ARM_MANIPULATION and flags & 128:
                self._aux_steppers = attach_arm_fall_models(self.stepper, self.blob, n_envs, self.device_index)
            elif flags & 16:
                self._aux_steppers = (attach_ragdoll_model(self.stepper, n_envs, self.device_index),)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _fresh_reset(self, mask, s):
        """FeedingEnv.reset for the envs selected by mask (None = all), in place on the device: env i of episode e is
        sampled from episode_seed(seed, e, env_offset) + i -- a function of the GLOBAL env index and the episode only,
        i.e. independent of the number of GPUs the batch is spread over -- and settled for 25 substeps"""
        from .model import compiler as L
        # FeedingEnv.reset: 25 steps for the food to drop; DressingEnv.reset: 50 for the garment; ScratchItchEnv.reset has no settle loop
        settle = {L.TASK_FEEDING: SETTLE_STEPS, L.TASK_DRESSING: DRESSING_SETTLE_STEPS, L.TASK_DRINKING: DRINKING_SETTLE_STEPS}.get(self.blob.task_kind, 0)
        self.stepper.reset(mask, None, episode_seed(self.seed, self._episode, self.env_offset), impairment=self.impairment,
                           settle_substeps=settle, stream=s)
        if settle > 0:
            # a sampled start deep inside the table / the bowl comes out of the settle blown up (a few in a thousand for the Panda, DESIGN 2:
            # no penetration-recovery clamp): such environments are drawn once more, from the seeds of a later "episode"
            st = self.stepper.state_tensor()
            fl = st[:, :self.blob.h['S_ENV']]                  # joint angles ... human frames, tremor words: the float part of a record (the env / task words hold integers)
            bad = ~torch.isfinite(fl).all(dim=1) | (st[:, self.blob.h['S_QD']:self.blob.h['S_QD'] + self.blob.ndof].abs().amax(dim=1) > 1.0e3)
            if mask is not None:
                bad &= mask.bool()
            if bool(bad.any()):
                self.stepper.reset(bad.to(torch.uint8), None, episode_seed(self.seed, self._episode + (1 << 20), self.env_offset), impairment=self.impairment,
                                   settle_substeps=settle, stream=s)
                self.start_states_redrawn += int(bad.sum())
        self._episode += 1

    def set_pool(self, states, cloth=None):
        """Use the given post-reset states (float32 [pool_size, state_words]; with a garment each for models with a cloth) as the reset
        pool instead of generating one -- e.g. a workload-specific pool (bench.py --workload wiping) or states sampled elsewhere."""
        assert states.dtype == np.float32 and states.shape == (self.pool_size, self.blob.state_words)
        self.pool_host = np.ascontiguousarray(states)
        self.pool = torch.from_numpy(self.pool_host).to(self.device)
        if cloth is not None:
            self.cloth_pool_host = np.ascontiguousarray(cloth)
            self.cloth_pool = torch.from_numpy(self.cloth_pool_host).to(self.device)
            self.stepper.set_cloth_pool(self.cloth_pool)

    def reset(self, env_offset=0):
        """env_offset: global index of this shard's first env (multi-GPU sharding keeps the
        env -> initial state mapping independent of the GPU count)."""
        self.env_offset, self._t = env_offset, 0
        s = self._stream()
        if self.reset_mode == 'device':
            self._fresh_reset(None, s)
            # rescue pool for environments the non-finite guard ends mid-episode (step()): the first start states of this batch
            k = min(self.n_envs, 64 if self.stepper.cloth_nodes() > 0 else 256)
            self._rescue = self.stepper.state_tensor()[:k].clone()
            if self.stepper.cloth_nodes() > 0:
                self._rescue_cloth = self.stepper.cloth_tensor()[:k].clone()
                self.stepper.set_cloth_pool(self._rescue_cloth)
        else:
            if self.pool is None:
                self.pool_host = build_reset_pool(self.blob, self.pool_size, self.seed, self.device_index, self.impairment,
                                                  sampler='host' if self.reset_mode == 'host' else 'device')
                if isinstance(self.pool_host, tuple):         # models with a cloth: (states, garments)
                    self.pool_host, self.cloth_pool_host = self.pool_host
                    self.cloth_pool = torch.from_numpy(self.cloth_pool_host).to(self.device)
                    self.stepper.set_cloth_pool(self.cloth_pool)
                self.pool = torch.from_numpy(self.pool_host).to(self.device)
                if self.pool_refresh > 0 and self._refresher is None:
                    self._refresher = PoolRefresher(self._model_name, self.blob.is_coop, self.seed, min(self.pool_refresh, self.pool_size), self.device_index, self.impairment,
                                                    'host' if self.reset_mode == 'host' else 'device', sync=self.pool_refresh_sync)
            idx = pool_indices(env_offset, self.n_envs, self.pool_size)
            self.stepper.set_state(self.pool_host[idx])
            if getattr(self, 'cloth_pool_host', None) is not None:
                self.stepper.set_cloth(self.cloth_pool_host[idx])
            self.stepper.set_env_offset(env_offset)       # later episodes draw from the pool by GLOBAL env index too
        self.stepper.observe_dev(self.obs, s)
        return self.obs

    def step(self, actions, obs_out=None):
        """obs_out: optional [n_envs, obs_dim] float32 device tensor that receives the observations instead of self.obs (double
        buffering for an overlapped all-gather, shard.ObsGatherer); it becomes self.obs"""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.shape == (self.n_envs, self.act_dim) and actions.is_contiguous()
        s = self._stream()
        if obs_out is not None:
            assert obs_out.is_cuda and obs_out.dtype == torch.float32 and obs_out.shape == (self.n_envs, self.obs_dim) and obs_out.is_contiguous()
            self.obs = obs_out
        self.stepper.step_dev(actions, self.obs, self.reward, self.done, self.info, s)
        self._t += 1
        if self.auto_reset:
            boundary = self._t % self.episode_len == 0      # lock-stepped batch: every env is done (feeding.py:37)
            if self.reset_mode != 'device':
                if boundary and self._refresher is not None:
                    self._swap_in_fresh_states()
                self.stepper.reset_done(self.pool, self.pool_size, self.done, s)
            elif boundary:
                self._fresh_reset(self.done, s)
            else:
                # an environment ended by the non-finite guard mid-episode (done = 1, zeroed outputs): replaced at once from the rescue pool
                # (the first states this batch was started from), joining the batch at its current iteration so that it reaches the
                # 200-step boundary -- where the masked agx_reset samples every environment anew -- together with the others.  No host
                # round trip: the kernel does nothing for environments that are not done.
                self.stepper.reset_done(self._rescue, len(self._rescue), self.done, s, iteration=self._t % self.episode_len)
            if not boundary:
                # ... and the row of such an environment becomes the first observation of its new episode, as at a boundary
                # (also the short episodes of workload-specific pools, bench.py --workload wiping); other rows are untouched
                if self.keep_terminal_obs:
                    self.terminal_obs = self.obs.clone()     # (before the masked re-observe overwrites the rows that ended: guard-ended environments, short-episode pools)
                self.stepper.observe_dev(self.obs, s, mask=self.done)
            if boundary:
                # vector-env convention: the observation returned with done is the first one of the new episode
                # (`return self._get_obs()` of reset(), feeding.py:182); the last one of the old episode is kept aside
                self.terminal_obs = self.obs.clone()
                self.stepper.observe_dev(self.obs, s)
        elif self.keep_terminal_obs:
            self.terminal_obs = self.obs
        return self.obs, self.reward, self.done, self.info

    def _swap_in_fresh_states(self):
        """batches the refresher has ready replace the oldest pool entries (round robin), states and -- for models with a cloth -- garments"""
        for b in self._refresher.poll():
            states, cloth = b if isinstance(b, tuple) else (b, None)
            k = len(states)
            idx = (self._refresh_cursor + np.arange(k)) % self.pool_size
            self.pool_host[idx] = states
            self.pool[torch.from_numpy(idx).to(self.device)] = torch.from_numpy(states).to(self.device)
            if cloth is not None:
                self.cloth_pool_host[idx] = cloth
                self.cloth_pool[torch.from_numpy(idx).to(self.device)] = torch.from_numpy(cloth).to(self.device)
            self._refresh_cursor = int((self._refresh_cursor + k) % self.pool_size)
            self.pool_refreshed += k

    def close(self):
        if self._refresher is not None:
            self._refresher.close()
        self.stepper.close()
        for st in self._aux_steppers:
            st.close()
        self._aux_steppers = ()


class FeedingJacoVecEnv(AssistiveVecEnv):
    model = 'feeding_jaco'


class FeedingSawyerVecEnv(AssistiveVecEnv):
    """FeedingSawyer-v1 (feeding_envs.py:25-27): a free-standing robot -- its base pose search runs in the device-side reset generator like
    the IK restarts of the mounted arms (reset='pool' / 'device'); reset='host': the numpy sampler"""
    model = 'feeding_sawyer'

    def __init__(self, n_envs, **kw):
        kw.setdefault('reset', 'pool')
        super().__init__(n_envs, **kw)


class FeedingStretchVecEnv(FeedingSawyerVecEnv):
    """FeedingStretch-v1 (feeding_envs.py:33-35): the mobile manipulator on its own wheels -- 5 actions (two wheels, lift, telescoping arm,
    wrist yaw: stretch.py:9-11,51-53), 21 observations (the wheel angles are left out, feeding.py:90-92).  Start states come from the
    device-side reset generator (its branch for a robot on wheels: env.py:282-293 has no IK) like those of the other feeding robots
    (reset='pool' / 'device'); reset='host': the numpy sampler."""
    model = 'feeding_stretch'


class FeedingBaxterVecEnv(FeedingSawyerVecEnv):
    """FeedingBaxter-v1 (feeding_envs.py:21-23): Baxter's right arm"""
    model = 'feeding_baxter'


class DrinkingJacoVecEnv(AssistiveVecEnv):
    """DrinkingJaco-v1 (drinking_envs.py:29-31): the wheelchair-mounted Jaco brings a cup with 64 water particles to the person's mouth.
    The water lives next to the state record like a garment (float32 [n, 2, 64, 3]); the pool holds (state, water) pairs sampled and settled
    on the device (agx_reset: IK restarts, the grid above the cup, 50 settle steps); reset='device': new ones every episode."""
    model = 'drinking_jaco'


class DrinkingPandaVecEnv(DrinkingJacoVecEnv):
    model = 'drinking_panda'


class DrinkingSawyerVecEnv(DrinkingJacoVecEnv):
    """DrinkingSawyer-v1 (drinking_envs.py:25-27): a free-standing robot, placed by the base pose search of the device-side reset generator
    (start pose and the mouth must be reachable, the mouth with the start orientation is the further goal: drinking.py:143)"""
    model = 'drinking_sawyer'


class DrinkingBaxterVecEnv(DrinkingJacoVecEnv):
    model = 'drinking_baxter'


class DrinkingPR2VecEnv(DrinkingJacoVecEnv):
    model = 'drinking_pr2'


class DrinkingStretchVecEnv(DrinkingJacoVecEnv):
    """DrinkingStretch-v1 (drinking_envs.py:33-35): the mobile manipulator, 5 actions, 21 observations"""
    model = 'drinking_stretch'


class FeedingPandaVecEnv(AssistiveVecEnv):
    """FeedingPanda-v1: the feeding kernels, oracle and device-side reset generator driven by the Panda's model blob."""
    model = 'feeding_panda'


class BedBathingSawyerVecEnv(AssistiveVecEnv):
    """BASELINE config 3.  reset='pool': a pool of post-reset states sampled once on the device -- the rag doll of a second model dropped onto
    the bed and settled for 100 steps, then the base pose search and the targets (agx_sample_reset with the rag-doll model attached);
    reset='device': every episode of every environment starts from a newly sampled and settled human (the settle of 4096 rag dolls takes
    about 1.7 s: DESIGN 8); reset='host': the numpy sampler around the device settle."""
    model = 'bed_bathing_sawyer'

    def __init__(self, n_envs, **kw):
        kw.setdefault('reset', 'pool')
        super().__init__(n_envs, **kw)


class ScratchItchPR2VecEnv(AssistiveVecEnv):
    """ScratchItchPR2-v1; coop=True = ScratchItchPR2Human-v1 (BASELINE config 4: 7 + 10 actions, 30 + 34 observations per env).
    reset='pool': a pool sampled once by the device-side reset generator (human, target, base pose search of position_robot_toc with its 50
    candidate poses one per lane, collision rejection); reset='device': new states every episode; reset='host': the numpy sampler."""
    model = 'scratch_itch_pr2'

    def __init__(self, n_envs, **kw):
        kw.setdefault('reset', 'pool')
        super().__init__(n_envs, **kw)
        assert self.reset_mode != 'device' or self.blob.has_reset_generator, 'no device-side reset generator for this model (the Sawyer needs the pedestal guard of the host sampler): use a pool'


class ScratchItchPR2HumanVecEnv(ScratchItchPR2VecEnv):
    coop = True


class ScratchItchJacoVecEnv(AssistiveVecEnv):
    """ScratchItchJaco-v1 (the reference's default environment): the scratch-itch kernels with the wheelchair-mounted Jaco's model blob;
    its resets come from the device-side reset generator (reset='pool': a pool sampled once; reset='device': fresh states every episode;
    reset='host': the numpy sampler of host/reset_scratch.py)"""
    model = 'scratch_itch_jaco'


class ScratchItchPandaVecEnv(ScratchItchJacoVecEnv):
    model = 'scratch_itch_panda'


class ScratchItchSawyerVecEnv(ScratchItchPR2VecEnv):
    model = 'scratch_itch_sawyer'


class ArmManipulationSawyerVecEnv(AssistiveVecEnv):
    """ArmManipulationSawyer-v1 (arm_manipulation_envs.py:23-25): 14 actions per env (the arm joints twice, robot.py:16), 45 observations."""
    model = 'arm_manipulation_sawyer'

    def __init__(self, n_envs, **kw):
        kw.setdefault('reset', 'pool')
        super().__init__(n_envs, **kw)
        assert self.reset_mode != 'device' or self.blob.has_reset_generator


class ArmManipulationSawyerHumanVecEnv(ArmManipulationSawyerVecEnv):
    coop = True


def _vec_flavour(base_cls, name, model_name):
    cls = type(name, (base_cls,), {'model': model_name, '__doc__': '%s: %s with another robot\'s model blob' % (name, base_cls.__name__)})
    globals()[name] = cls
    return cls


for _r in ('jaco', 'panda', 'pr2', 'baxter'):
    _vec_flavour(BedBathingSawyerVecEnv, 'BedBathing%sVecEnv' % {'pr2': 'PR2'}.get(_r, _r.capitalize()), 'bed_bathing_' + _r)
_vec_flavour(ScratchItchPR2VecEnv, 'ScratchItchBaxterVecEnv', 'scratch_itch_baxter')
_vec_flavour(ScratchItchPR2VecEnv, 'ScratchItchStretchVecEnv', 'scratch_itch_stretch')      # mobile: 5 actions, 26 observations
_vec_flavour(BedBathingSawyerVecEnv, 'BedBathingStretchVecEnv', 'bed_bathing_stretch')      # mobile: 5 actions, 20 observations
_vec_flavour(FeedingSawyerVecEnv, 'FeedingPR2VecEnv', 'feeding_pr2')


_vec_flavour(ArmManipulationSawyerVecEnv, 'ArmManipulationJacoVecEnv', 'arm_manipulation_jaco')
_vec_flavour(ArmManipulationSawyerVecEnv, 'ArmManipulationPandaVecEnv', 'arm_manipulation_panda')
_vec_flavour(ArmManipulationSawyerVecEnv, 'ArmManipulationPR2VecEnv', 'arm_manipulation_pr2')          # two arms, two tools
_vec_flavour(ArmManipulationSawyerVecEnv, 'ArmManipulationBaxterVecEnv', 'arm_manipulation_baxter')


class DressingBaxterVecEnv(AssistiveVecEnv):
    """BASELINE config 5: DressingBaxter-v1 (dressing_envs.py:19-21).  Every environment carries a garment of 3,966 nodes next to its
    state record.  reset='pool': (state, settled garment) pairs generated once on the device (sampling incl. the base pose search, the
    garment at the end effector, the 50-step settle: agx_reset); reset='device': every episode of every environment starts from a newly
    sampled and settled state; reset='host': the numpy sampler of host/reset_dressing.py around the device settle."""
    model = 'dressing_baxter'

    def __init__(self, n_envs, **kw):
        kw.setdefault('reset', 'pool')
        kw.setdefault('pool_size', 64)
        super().__init__(n_envs, **kw)
        assert self.reset_mode != 'device' or self.blob.has_reset_generator, 'no device-side reset generator for this model (the Sawyer needs the pedestal guard of the host sampler): use a pool'


class DressingBaxterHumanVecEnv(DressingBaxterVecEnv):
    coop = True


for _r in ('sawyer', 'jaco', 'panda', 'pr2', 'stretch'):      # (stretch: mobile, 5 actions, 20 observations)
    _vec_flavour(DressingBaxterVecEnv, 'Dressing%sVecEnv' % {'pr2': 'PR2'}.get(_r, _r.capitalize()), 'dressing_' + _r)
