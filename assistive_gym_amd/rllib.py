"""RLlib adapter: the batched stepper as ONE `ray.rllib.env.VectorEnv`, so that a rollout worker steps thousands of
environments with a few kernel launches instead of `num_workers = cpu_count()` processes with one PyBullet client each
(assistive_gym/learn.py:9-37,71-94).  Usage with the reference's training script left unchanged:

    from ray.tune.registry import register_env
    register_env('assistive_gym:FeedingJaco-v1', lambda cfg: AgxVectorEnv('FeedingJaco-v1', cfg.get('num_envs', 1024)))

RLlib's VectorEnv contract (ray 1.x): vector_reset() -> [obs], reset_at(i) -> obs, vector_step(actions) ->
([obs], [reward], [done], [info]), get_unwrapped() -> [envs].  Episodes of the lock-stepped batch all end at step 200
(done = iteration >= 200 in every task's step()); the batch is then reset on the device and reset_at(i) returns the
first observation of env i's new episode.  ray is optional at import time."""
import numpy as np

try:
    from ray.rllib.env.vector_env import VectorEnv as _VectorEnv
except Exception:
    class _VectorEnv:                    # the attributes RLlib reads
        def __init__(self, observation_space, action_space, num_envs):
            self.observation_space, self.action_space, self.num_envs = observation_space, action_space, num_envs

def _models():
    """env id -> model blob of every single-agent environment that is built (assistive_gym_amd.envs.ENV_IDS)"""
    from . import envs
    return {k: c.model for k, c in envs.ENV_IDS.items() if not c.coop}


class _LazyInfos:
    """the per-environment info dicts of a step (`info` of <Task>Env.step: total_force_on_human, task_success, the four length entries),
    built only for the environments somebody indexes: a sampler that never looks at `info` does not pay for num_envs dictionaries per step"""

    def __init__(self, static, force, success):
        self._static, self._force, self._success = static, force, success

    def __len__(self):
        return len(self._force)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        return dict(self._static, total_force_on_human=float(self._force[i]), task_success=int(self._success[i]))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _Rows:
    """the rows of an [n, d] array as a sequence (what RLlib's sampler iterates: `dict(enumerate(obs))`), without building n array views up
    front -- 1.1 ms of host time per step at 4096 environments"""

    def __init__(self, a):
        self._a = a

    def __len__(self):
        return len(self._a)

    def __getitem__(self, i):
        return self._a[i]

    def __iter__(self):
        return iter(self._a)


class AgxVectorEnv(_VectorEnv):
    def __init__(self, env_id, num_envs, device=0, seed=1001, reuse_host_buffers=False, env_offset=0, **vec_kwargs):
        """env_offset: global index of this object's first environment -- several AgxVectorEnv objects (RLlib rollout workers: `worker_env`
        below) then step disjoint slices of ONE batch (pool draws and reset seeds go by global index, as across GPUs: shard.py).
        reuse_host_buffers (opt-in, default off): hand out VIEWS into two pinned host buffers used in turn instead of a fresh copy per step.
        The views are overwritten two steps later, so this is only for a consumer that is done with a step's observations / infos before
        the step after next -- NOT RLlib's collectors, which keep the ndarray references of a whole rollout fragment and stack them when
        the SampleBatch is built.  With the default every step's rows live in memory of their own, valid for ever."""
        from . import envs
        from .vec_env import AssistiveVecEnv
        self._reuse = bool(reuse_host_buffers)
        self._env_offset = int(env_offset)
        name = env_id.split(':')[-1]
        proto = envs.ENV_IDS[name]()                     # spaces and info keys of the scalar env
        assert not proto.coop, 'co-op (multi-agent) ids are batched by AgxMultiAgentBatchEnv (RLlib BaseEnv), not by a VectorEnv'
        super().__init__(proto.observation_space, proto.action_space, num_envs)
        self._info_static = {'action_robot_len': proto.action_robot_len, 'action_human_len': proto.action_human_len,
                             'obs_robot_len': proto.obs_robot_len, 'obs_human_len': proto.obs_human_len}
        vec_kwargs.setdefault('reset', 'pool')             # (reset='device': fresh states every episode, for the models with a device-side reset generator)
        self.vec = AssistiveVecEnv(num_envs, device=device, seed=seed, model=_models()[name], **vec_kwargs)
        self.vec.keep_terminal_obs = True                 # the last observation of every row that ends, boundary or not (vec_env.step)
        self._obs, self._host, self._pack = None, None, None

    def vector_reset(self):
        self._obs = _Rows(self.vec.reset(env_offset=self._env_offset).cpu().numpy())
        return self._obs

    def reset_at(self, index):
        # the batch was reset on the device at the episode boundary; an environment the non-finite guard ended mid-episode was replaced at
        # once (vec_env.step) and the first observation of its new episode was kept for this call
        return self._obs[index]

    def vector_step(self, actions):
        """Host cost per step at 4096 environments (round 5, tools/gpu_rllib_overhead.py): the actions arrive as a list of arrays (one
        np.concatenate, 0.5 ms; an [n, act_dim] array is taken as it is), ONE device-to-host transfer brings back terminal observations,
        rewards, done flags, two info columns and -- for the rows that ended -- the first observations of the new episodes, into a pinned
        buffer; what is handed out is ONE contiguous copy of it per step (0.1 ms at 4096 x 54 floats), float32 like the observation space:
        RLlib's collectors keep the row references until the SampleBatch is built, so the rows must never be overwritten
        (reuse_host_buffers=True hands out views into two alternating pinned buffers instead: see __init__)."""
        import time
        import torch
        n = self.num_envs
        t0 = time.perf_counter()
        if not isinstance(actions, np.ndarray):
            actions = np.concatenate(actions).reshape(n, -1)
        a = torch.as_tensor(np.ascontiguousarray(actions, dtype=np.float32), device=self.vec.device)
        t1 = time.perf_counter()
        obs, rew, done, info = self.vec.step(a)
        # rows whose episode ended: RLlib wants the LAST observation of the old episode here and the first of the new one from reset_at(); the
        # stepper has already put the new first observation into `obs` for those rows and kept the old one in vec.terminal_obs
        last = self.vec.terminal_obs
        od = obs.shape[1]
        if self._host is None:
            self._pack = torch.empty((n, 2 * od + 4), dtype=torch.float32, device=self.vec.device)
            self._host = [torch.empty((n, 2 * od + 4), dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self._flip = 0
        pk = self._pack
        pk[:, :od] = last; pk[:, od] = rew; pk[:, od + 1] = done; pk[:, od + 2:od + 4] = info[:, 0:2]; pk[:, od + 4:] = obs
        self._flip ^= 1
        host = self._host[self._flip]
        host.copy_(pk, non_blocking=True)
        t2 = time.perf_counter()
        torch.cuda.current_stream(self.vec.device).synchronize()
        t3 = time.perf_counter()
        self.host_seconds = getattr(self, 'host_seconds', np.zeros(4)) + np.array([t1 - t0, t2 - t1, t3 - t2, 0.0])      # actions to the device | enqueue | wait for the GPU | (results: added below)
        h = host.numpy()
        if not self._reuse:
            h = h.copy()                                  # memory of this step's own: rows, info columns and reset_at() observations are views of it
        dn = h[:, od + 1] != 0
        rows = _Rows(h[:, :od])
        # reset_at(i): the first observation of the new episode for the rows that ended, the current observation elsewhere
        self._obs = _Rows(np.where(dn[:, None], h[:, od + 4:], h[:, :od])) if dn.any() else rows
        out = rows, h[:, od].tolist(), dn.tolist(), _LazyInfos(self._info_static, h[:, od + 2], h[:, od + 3])
        self.host_seconds[3] += time.perf_counter() - t3
        return out

    def get_unwrapped(self):
        return []

    def close(self):
        self.vec.close()


def worker_env(env_id, cfg, default_envs=2048):
    """env creator for `register_env`: one AgxVectorEnv per RLlib rollout worker, each a disjoint slice of one batch.

        register_env('assistive_gym:FeedingJaco-v1', lambda cfg: worker_env('FeedingJaco-v1', cfg))
        config = {'num_workers': 2, 'env_config': {'num_envs': 2048}, ...}

    `vector_step` is synchronous (RLlib's VectorEnv contract): while a worker's sampler builds its batch the GPU would idle, and while the GPU steps
    the worker would.  With TWO workers of 2,048 environments on one GPU (the reference runs `num_workers = cpu_count()` of them, learn.py:26,72) one
    worker's kernels run while the other is in its Python: measured in tools/gpu_rllib_overhead.py (two threads / two processes).  cfg: RLlib's
    EnvContext (worker_index 1 ... num_workers; 0 = the local worker) or a plain dict."""
    n = int(cfg.get('num_envs', default_envs)) if hasattr(cfg, 'get') else default_envs
    w = max(0, int(getattr(cfg, 'worker_index', 0)) - 1)
    return AgxVectorEnv(env_id, n, device=int(cfg.get('device', 0)) if hasattr(cfg, 'get') else 0, seed=int(cfg.get('seed', 1001)) if hasattr(cfg, 'get') else 1001, env_offset=w * n)


try:
    from ray.rllib.env.base_env import BaseEnv as _BaseEnv
except Exception:
    class _BaseEnv:
        pass


class AgxMultiAgentBatchEnv(_BaseEnv):
    """The co-op ids (`<Task><Robot>Human-v1`: RLlib MultiAgentEnv in the reference, two policies 'robot' and 'human' --
    assistive_gym/learn.py:33-36, e.g. scratch_itch_envs.py:41-44 = BASELINE config 4) as ONE RLlib BaseEnv over the batched stepper: what
    `num_workers = cpu_count()` processes with one scalar multi-agent env each do in the reference.  RLlib's BaseEnv contract (ray 1.x):

        poll() -> (obs, rewards, dones, infos, off_policy_actions), each {env_id: {agent_id: value}} (dones also carry '__all__')
        send_actions({env_id: {agent_id: action}});   try_reset(env_id) -> {agent_id: obs};   get_unwrapped() -> []

    One env.step() of all environments and ONE device-to-host transfer per poll/send round (a single pinned buffer: both agents'
    observations, reward, done, two info columns).  Register it the way the reference registers its scalar multi-agent envs:

        register_env('assistive_gym:ScratchItchPR2Human-v1', lambda cfg: AgxMultiAgentBatchEnv('ScratchItchPR2Human-v1', cfg.get('num_envs', 1024)))
    """

    def __init__(self, env_id, num_envs, device=0, seed=1001, **vec_kwargs):
        from . import envs
        from .vec_env import AssistiveVecEnv
        name = env_id.split(':')[-1]
        cls = envs.ENV_IDS[name]
        assert cls.coop, 'single-agent ids go through AgxVectorEnv'
        proto = cls()
        self.num_envs = num_envs
        self.observation_space_robot, self.observation_space_human = proto.observation_space_robot, proto.observation_space_human
        self.action_space_robot, self.action_space_human = proto.action_space_robot, proto.action_space_human
        self.observation_space, self.action_space = proto.observation_space, proto.action_space
        self.nr, self.ar = proto.obs_robot_len, proto.action_robot_len
        self._info_static = {'action_robot_len': proto.action_robot_len, 'action_human_len': proto.action_human_len,
                             'obs_robot_len': proto.obs_robot_len, 'obs_human_len': proto.obs_human_len}
        vec_kwargs.setdefault('reset', 'pool')
        self.vec = AssistiveVecEnv(num_envs, device=device, seed=seed, model=cls.model, coop=True, **vec_kwargs)
        self._host = self._pack = None
        self._pending = None             # what the next poll() returns
        self.vec.keep_terminal_obs = True
        self._first = self._split(self.vec.reset().cpu().numpy().astype(np.float64))
        self._pending = self._after_reset(self._first)
        self._new_obs = {}               # env_id -> first observation of the episode that began at the last boundary (try_reset)

    @staticmethod
    def _after_reset(first):
        """what poll() returns for environments that have just been reset: as ray 1.x's _MultiAgentEnvState.reset leaves them -- a reward of None
        per agent, done False for both agents and '__all__', an empty info per agent (RLlib's sampler indexes dones[env_id]['__all__'] and
        infos[env_id] for every env_id that has an observation)"""
        return (dict(first), {i: {'robot': None, 'human': None} for i in first}, {i: {'robot': False, 'human': False, '__all__': False} for i in first},
                {i: {'robot': {}, 'human': {}} for i in first})

    def _split(self, obs):
        return {i: {'robot': obs[i, :self.nr], 'human': obs[i, self.nr:]} for i in range(self.num_envs)}

    def poll(self):
        obs, rew, done, info = self._pending
        self._pending = ({}, {}, {}, {})
        return obs, rew, done, info, {}

    def send_actions(self, action_dict):
        import torch
        n = self.num_envs
        a = np.zeros((n, self.vec.act_dim), dtype=np.float32)
        for i, d in action_dict.items():
            a[i, :self.ar] = d['robot']; a[i, self.ar:] = d['human']
        obs, rew, done, info = self.vec.step(torch.as_tensor(a, device=self.vec.device).contiguous())
        last = self.vec.terminal_obs                    # the last observation of every row that ended (boundary or not), the current one elsewhere
        od = last.shape[1]
        if self._host is None:
            self._pack = torch.empty((n, od + 4), dtype=torch.float32, device=self.vec.device)
            self._host = torch.empty((n, od + 4), dtype=torch.float32, pin_memory=True)
        self._pack[:, :od] = last; self._pack[:, od] = rew; self._pack[:, od + 1] = done.float(); self._pack[:, od + 2:od + 4] = info[:, 0:2]
        self._host.copy_(self._pack, non_blocking=True)
        torch.cuda.current_stream(self.vec.device).synchronize()
        h = self._host.numpy()
        o = self._split(h[:, :od].astype(np.float64))
        r = {i: {'robot': float(h[i, od]), 'human': float(h[i, od])} for i in range(n)}                       # both agents share the reward (feeding.py:41-43)
        d = {i: {'robot': bool(h[i, od + 1]), 'human': bool(h[i, od + 1]), '__all__': bool(h[i, od + 1])} for i in range(n)}
        inf = {i: {'robot': dict(self._info_static, total_force_on_human=float(h[i, od + 2]), task_success=int(h[i, od + 3]))} for i in range(n)}
        for i in range(n):
            inf[i]['human'] = inf[i]['robot']
        if h[:, od + 1].any():           # environments that ended (all of them at the 200-step boundary): their next observation is the new episode's first
            first = self._split(obs.cpu().numpy().astype(np.float64))
            self._new_obs = {i: first[i] for i in range(n) if h[i, od + 1]}
        self._pending = (o, r, d, inf)

    def try_reset(self, env_id=None):
        """RLlib calls this for an env whose '__all__' is done; the batch has already been reset on the device"""
        if env_id is None:
            return {i: self.try_reset(i) for i in range(self.num_envs)}
        return self._new_obs.pop(env_id, None) or self._first[env_id]

    def get_unwrapped(self):
        return []

    def stop(self):
        self.vec.close()

    close = stop


_AGENT = 'agent0'          # RLlib's _DUMMY_AGENT_ID: the agent key of a single-agent env behind the BaseEnv interface


class AgxPipelinedBatchEnv(_BaseEnv):
    """The single-agent ids as an RLlib BaseEnv with TWO half-batches in flight.

    `vector_step` of a VectorEnv is synchronous: every step starts on an idle GPU and the sampler's own work (policy forward, collectors)
    starts on an idle host -- 334 k of the 554 k env-steps/s the device-resident loop reaches (profiles/r05).  RLlib's BaseEnv contract is
    asynchronous by design (`poll()` returns whichever sub-environments have something to report, `send_actions()` takes actions for exactly
    those; the sampler loop of ray 1.x, _env_runner, alternates the two), so this adapter splits the batch into halves A = [0, n/2) and
    B = [n/2, n): poll() hands out the finished step of one half while the kernels of the other half run, send_actions() enqueues that
    half's next step and returns without waiting.  Each half is a handle of its own (same pool, global env indices: the trajectories are those
    of one n-environment batch), stepped on its own stream, with its own pinned host buffer; what poll() returns is a copy (RLlib keeps the
    references until the SampleBatch is built).  Episode ends: as AgxVectorEnv -- the observation returned with done is the LAST one of the
    old episode, try_reset(env_id) the first one of the new episode.

        register_env('assistive_gym:FeedingJaco-v1', lambda cfg: AgxPipelinedBatchEnv('FeedingJaco-v1', cfg.get('num_envs', 4096)))
    """

    def __init__(self, env_id, num_envs, device=0, seed=1001, **vec_kwargs):
        import torch
        from . import envs
        from .vec_env import AssistiveVecEnv
        name = env_id.split(':')[-1]
        proto = envs.ENV_IDS[name]()
        assert not proto.coop and num_envs % 2 == 0
        self.num_envs, self.observation_space, self.action_space = num_envs, proto.observation_space, proto.action_space
        self._info_static = {'action_robot_len': proto.action_robot_len, 'action_human_len': proto.action_human_len,
                             'obs_robot_len': proto.obs_robot_len, 'obs_human_len': proto.obs_human_len}
        vec_kwargs.setdefault('reset', 'pool')
        n2 = num_envs // 2
        self.n2, self.torch = n2, torch
        self.halves, self.streams, self.events, self.host, self.pack, self.act = [], [], [], [], [], []
        for h in range(2):
            v = AssistiveVecEnv(n2, device=device, seed=seed, model=_models()[name], **vec_kwargs)
            v.keep_terminal_obs = True
            self.halves.append(v); self.streams.append(torch.cuda.Stream(device=v.device)); self.events.append(torch.cuda.Event())
        od = self.halves[0].obs_dim
        self.od = od
        for h in range(2):
            v = self.halves[h]
            if h == 1:
                v.pool, v.pool_host = self.halves[0].pool, self.halves[0].pool_host          # one pool for both halves (drawn by GLOBAL env index)
            with torch.cuda.stream(self.streams[h]):
                first = v.reset(env_offset=h * n2)
                self.pack.append(torch.empty((n2, 2 * od + 4), dtype=torch.float32, device=v.device))
                self.host.append(torch.empty((n2, 2 * od + 4), dtype=torch.float32, pin_memory=True))
                self.act.append(torch.empty((n2, v.act_dim), dtype=torch.float32, device=v.device))
                self.host[h][:, :od].copy_(first, non_blocking=True)
                self.events[h].record(self.streams[h])
        self._state = ['reset', 'reset']      # what the next poll of a half returns: its reset observations, or a step's results
        self._turn = 0
        self._new_obs = {}
        self._ids = [np.arange(0, n2), np.arange(n2, num_envs)]

    def poll(self):
        h = self._turn
        self.events[h].synchronize()
        a = self.host[h].numpy().copy()
        od, ids = self.od, self._ids[h].tolist()
        if self._state[h] == 'reset':
            obs = {i: {_AGENT: row} for i, row in zip(ids, a[:, :od])}
            return (obs, {i: {_AGENT: None} for i in ids}, {i: {_AGENT: False, '__all__': False} for i in ids}, {i: {_AGENT: {}} for i in ids}, {})
        dn = (a[:, od + 1] != 0).tolist()
        obs = {i: {_AGENT: row} for i, row in zip(ids, a[:, :od])}
        rew = {i: {_AGENT: r} for i, r in zip(ids, a[:, od].tolist())}
        done = {i: {_AGENT: d, '__all__': d} for i, d in zip(ids, dn)}
        info = _LazyAgentInfos(ids[0], self._info_static, a[:, od + 2], a[:, od + 3])
        if any(dn):
            first = a[:, od + 4:]
            for k, d in enumerate(dn):
                if d:
                    self._new_obs[ids[k]] = {_AGENT: first[k]}
        return obs, rew, done, info, {}

    def send_actions(self, action_dict):
        """actions for the environments of the half the last poll() reported; enqueues that half's next step and returns"""
        torch = self.torch
        h = self._turn
        v, n2, od = self.halves[h], self.n2, self.od
        lo = h * n2
        a = np.empty((n2, v.act_dim), dtype=np.float32)
        if len(action_dict) == n2:
            for k in range(n2):
                a[k] = action_dict[lo + k][_AGENT]
        else:                                  # (a sampler that skipped some environments: they repeat a zero action)
            a[:] = 0
            for i, d in action_dict.items():
                a[i - lo] = d[_AGENT]
        with torch.cuda.stream(self.streams[h]):
            self.act[h].copy_(torch.from_numpy(a), non_blocking=True)
            obs, rew, done, info = v.step(self.act[h])
            pk = self.pack[h]
            pk[:, :od] = v.terminal_obs; pk[:, od] = rew; pk[:, od + 1] = done; pk[:, od + 2:od + 4] = info[:, 0:2]; pk[:, od + 4:] = obs
            self.host[h].copy_(pk, non_blocking=True)
            self.events[h].record(self.streams[h])
        self._state[h] = 'step'
        self._turn ^= 1

    def try_reset(self, env_id=None):
        if env_id is None:
            return {i: self.try_reset(i) for i in range(self.num_envs)}
        return self._new_obs.pop(env_id, None)

    def get_unwrapped(self):
        return []

    def stop(self):
        for v in self.halves:
            v.close()

    close = stop


class _LazyAgentInfos:
    """{env_id: {'agent0': info}} of a half-batch, built per environment on access (a sampler that never reads `info` pays nothing)"""

    def __init__(self, first_id, static, force, success):
        self._lo, self._static, self._force, self._success = first_id, static, force, success

    def __len__(self):
        return len(self._force)

    def __contains__(self, i):
        return self._lo <= i < self._lo + len(self._force)

    def __getitem__(self, i):
        k = i - self._lo
        if not 0 <= k < len(self._force):
            raise KeyError(i)
        return {_AGENT: dict(self._static, total_force_on_human=float(self._force[k]), task_success=int(self._success[k]))}

    def get(self, i, default=None):
        return self[i] if i in self else default

    def keys(self):
        return range(self._lo, self._lo + len(self._force))

    def items(self):
        return ((i, self[i]) for i in self.keys())

    def __iter__(self):
        return iter(self.keys())
