"""`learn.py` drops in unchanged: the reference's own make_env (assistive_gym/learn.py:61-69) is executed against
the drop-in `assistive_gym` package with stub `gym` / `ray` modules (neither is installed here).  The reference source is
read from /root/reference at test time (never copied into the repo); the test skips where the reference is absent."""
import ast
import importlib
import os
import sys
import types

import pytest

REF_LEARN = '/root/reference/assistive_gym/learn.py'


class _TimeLimit:
    """what gym.make wraps a registered env in when max_episode_steps is given"""

    def __init__(self, env, max_episode_steps):
        self.env, self._max_episode_steps = env, max_episode_steps

    def __getattr__(self, name):
        return getattr(self.env, name)


def _stub_gym():
    gym = types.ModuleType('gym')
    reg = {}

    def register(id, entry_point, max_episode_steps=None, **kw):
        if id in reg:
            raise RuntimeError('Cannot re-register id: ' + id)
        reg[id] = (entry_point, max_episode_steps)

    def make(name):
        if ':' in name:                                  # gym imports the module before the colon, then looks the id up
            mod, name = name.split(':')
            importlib.import_module(mod)
        if name not in reg:
            raise KeyError('No registered env with id: ' + name)
        entry, steps = reg[name]
        mod, cls = entry.split(':')
        return _TimeLimit(getattr(importlib.import_module(mod), cls)(), steps)
    gym.make, gym.registry = make, reg
    envs = types.ModuleType('gym.envs'); registration = types.ModuleType('gym.envs.registration')
    registration.register = register
    envs.registration = registration; gym.envs = envs
    return gym, {'gym': gym, 'gym.envs': envs, 'gym.envs.registration': registration}


def _stub_ray():
    ray = types.ModuleType('ray'); tune = types.ModuleType('ray.tune'); registry = types.ModuleType('ray.tune.registry')
    creators = {}
    registry.register_env = lambda name, creator: creators.__setitem__(name, creator)
    ray.tune, tune.registry = tune, registry
    return creators, {'ray': ray, 'ray.tune': tune, 'ray.tune.registry': registry}


@pytest.fixture()
def shimmed(monkeypatch):
    from assistive_gym_amd import shim
    for m in [k for k in sys.modules if k == 'assistive_gym' or k.startswith('assistive_gym.')]:
        monkeypatch.delitem(sys.modules, m)
    gym, gmods = _stub_gym()
    creators, rmods = _stub_ray()
    for k, v in {**gmods, **rmods}.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.syspath_prepend(shim.install())
    yield gym, creators
    for m in [k for k in sys.modules if k == 'assistive_gym' or k.startswith('assistive_gym.')]:
        sys.modules.pop(m, None)


def _reference_make_env(gym):
    if not os.path.exists(REF_LEARN):
        pytest.skip('reference not on this box')
    tree = ast.parse(open(REF_LEARN).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'make_env'][0]
    ns = {'gym': gym, 'importlib': importlib}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF_LEARN, 'exec'), ns)
    return ns['make_env']


def _built_ids(coop):
    from assistive_gym_amd.envs import ENV_IDS
    return sorted(k for k in ENV_IDS if k.endswith('Human-v1') == coop)


@pytest.mark.parametrize('env_name', _built_ids(False))
def test_reference_make_env_single_agent(shimmed, env_name):
    gym, _ = shimmed
    make_env = _reference_make_env(gym)
    env = make_env(env_name, coop=False, seed=7)
    assert type(env.env).__name__ == env_name.split('-')[0] + 'Env' and env._max_episode_steps == 200
    n_act = 14 if env_name.startswith('ArmManipulation') else 5 if 'Stretch' in env_name else 7         # robot_arm = 'both' on the single-arm Sawyer (robot.py:16); Stretch: stretch.py:9-11
    assert env.action_space.shape == (n_act,) and env.observation_space.shape[0] in (20, 21, 24, 25, 26, 30, 45)
    assert env.action_robot_len == n_act and len(env.robot.controllable_joint_indices) == n_act      # what learn.py / env_viewer.py read
    env.disconnect()


@pytest.mark.parametrize('env_name', _built_ids(True))
def test_reference_make_env_coop(shimmed, env_name):
    gym, creators = shimmed
    make_env = _reference_make_env(gym)
    env = make_env(env_name, coop=True, seed=7)
    assert type(env).__name__ == env_name.split('-')[0] + 'Env' and env.human.controllable
    # setup_config (learn.py:31-36) reads these four spaces to build the two policies
    assert env.observation_space_robot.shape[0] + env.observation_space_human.shape[0] == env.observation_space.shape[0]
    assert env.action_space_robot.shape[0] + env.action_space_human.shape[0] == env.action_space.shape[0]
    # and RLlib finds the env under the id the reference registers it with (feeding_envs.py:67, bed_bathing_envs.py:61)
    assert 'assistive_gym:' + env_name in creators
    assert type(creators['assistive_gym:' + env_name]({})).__name__ == type(env).__name__
    env.disconnect()


def test_unbuilt_env_id_fails_like_gym(shimmed):
    gym, _ = shimmed
    with pytest.raises(KeyError):
        gym.make('assistive_gym:FeedingJacoMesh-v1')          # the mesh (SMPL-X) humans are out of scope; ArmManipulationStretch raises in the reference itself


def test_vector_env_adapter_surface():
    """the RLlib VectorEnv adapter imports without ray and exposes the contract's methods"""
    from assistive_gym_amd.rllib import AgxVectorEnv
    for m in ('vector_reset', 'reset_at', 'vector_step', 'get_unwrapped'):
        assert callable(getattr(AgxVectorEnv, m))


@pytest.mark.gpu
@pytest.mark.parametrize('env_id', ['FeedingJaco-v1', 'assistive_gym:BedBathingSawyer-v1'])
def test_vector_env_adapter_steps_on_the_gpu(env_id):
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd.rllib import AgxVectorEnv
    n = 32
    env = AgxVectorEnv(env_id, n, pool_size=8)
    obs = env.vector_reset()
    assert len(obs) == n and obs[0].shape == env.observation_space.shape
    rng = np.random.RandomState(0)
    for k in range(200):
        obs, rew, done, infos = env.vector_step([rng.uniform(-1, 1, 7) for _ in range(n)])
        assert len(obs) == len(rew) == len(done) == len(infos) == n and all(d == (k == 199) for d in done)
    assert set(infos[0]) == {'total_force_on_human', 'task_success', 'action_robot_len', 'action_human_len', 'obs_robot_len', 'obs_human_len'}
    first = env.reset_at(3)                               # first observation of the next episode, already on the host
    assert first.shape == env.observation_space.shape and np.isfinite(first).all() and not np.array_equal(first, obs[3])
    env.close()


@pytest.mark.gpu
def test_vector_env_rows_are_never_overwritten_and_workers_slice_one_batch():
    """ RLlib's collectors keep the observation / info references of a whole rollout fragment and stack them when the SampleBatch is
    built: what vector_step hands out must never change afterwards -- the rows of step k are compared with copies taken at step k after 6 more steps
    (the opt-in reuse_host_buffers=True, two alternating pinned buffers, DOES overwrite them: the contract its docstring states).  And: two
    AgxVectorEnv objects of n / 2 environments with env_offset 0 / n/2 (rllib.worker_env: two rollout workers) step the environments of ONE
    n-environment batch: same first observations, same trajectory under the same actions."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd.rllib import AgxVectorEnv, AgxPipelinedBatchEnv, worker_env
    n = 16
    rng = np.random.RandomState(0)
    acts = [rng.uniform(-1, 1, (n, 7)).astype(np.float32) for _ in range(8)]
    for reuse in (False, True):
        env = AgxVectorEnv('FeedingJaco-v1', n, pool_size=8, reuse_host_buffers=reuse)
        env.vector_reset()
        kept, infos0 = [], None
        for k in range(8):
            obs, rew, done, infos = env.vector_step(list(acts[k]))
            kept.append(([obs[i] for i in range(n)], [obs[i].copy() for i in range(n)]))
            if k == 0:
                infos0 = (infos, [dict(infos[i]) for i in range(n)])
        same = all(np.array_equal(r, c) for rows, copies in kept[:2] for r, c in zip(rows, copies))
        assert same == (not reuse)
        if not reuse:
            assert all(infos0[0][i] == infos0[1][i] for i in range(n))
            whole = kept
        env.close()
    halves = [AgxVectorEnv('FeedingJaco-v1', n // 2, pool_size=8, env_offset=h * (n // 2)) for h in range(2)]
    w = worker_env('FeedingJaco-v1', {'num_envs': n // 2})
    assert w.num_envs == n // 2 and w._env_offset == 0
    w.close()
    first = [h.vector_reset() for h in halves]
    for k in range(3):
        outs = [halves[h].vector_step(list(acts[k][h * (n // 2):(h + 1) * (n // 2)])) for h in range(2)]
        for h in range(2):
            for i in range(n // 2):
                assert np.array_equal(outs[h][0][i], whole[k][1][h * (n // 2) + i]), (k, h, i)
    for h in halves:
        h.close()
    # the asynchronous two-half BaseEnv: the same batch again, half by half
    p = AgxPipelinedBatchEnv('FeedingJaco-v1', n, pool_size=8)
    for k in range(3):
        for h in range(2):
            obs, rew, done, info, _ = p.poll()
            ids = sorted(obs)
            assert ids == list(range(h * (n // 2), (h + 1) * (n // 2))) and set(obs[ids[0]]) == {'agent0'}
            if k > 0:
                assert all(np.array_equal(obs[i]['agent0'], whole[k - 1][1][i]) for i in ids) and not done[ids[0]]['__all__'] and 'task_success' in info[ids[0]]['agent0']
            else:
                assert all(rew[i]['agent0'] is None for i in ids)
            p.send_actions({i: {'agent0': acts[k][i]} for i in ids})
    p.stop()


def test_multi_agent_batch_adapter_surface():
    """the co-op batch adapter (RLlib BaseEnv contract) imports without ray"""
    from assistive_gym_amd.rllib import AgxMultiAgentBatchEnv
    for m in ('poll', 'send_actions', 'try_reset', 'get_unwrapped', 'stop'):
        assert callable(getattr(AgxMultiAgentBatchEnv, m))


@pytest.mark.gpu
def test_multi_agent_batch_adapter_runs_config4_batched():
    """BASELINE config 4's environment (ScratchItchPR2Human-v1: two policies, 7 + 10 actions, 30 + 34 observations) through the co-op batch
    adapter: one poll / send_actions round = one env.step() of the whole batch; per-agent dictionaries as the reference's MultiAgentEnv
    returns them (scratch_itch.py:38-40); the same numbers as the scalar multi-agent env from the same state"""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd.rllib import AgxMultiAgentBatchEnv
    from assistive_gym_amd.envs import ENV_IDS
    n = 16
    env = AgxMultiAgentBatchEnv('assistive_gym:ScratchItchPR2Human-v1', n, pool_size=8)
    obs, rew, done, info, off = env.poll()
    assert sorted(obs) == list(range(n)) and obs[0]['robot'].shape == (30,) and obs[0]['human'].shape == (34,) and off == {}
    # after a reset: what ray 1.x's _MultiAgentEnvState.reset leaves (the sampler indexes dones[env_id]['__all__'] and infos[env_id] for every observed env)
    assert all(rew[i] == {'robot': None, 'human': None} and done[i] == {'robot': False, 'human': False, '__all__': False} and info[i] == {'robot': {}, 'human': {}} for i in range(n))
    scalar = ENV_IDS['ScratchItchPR2Human-v1']()
    scalar.set_state(env.vec.stepper.get_state()[5])
    rng = np.random.RandomState(0)
    for k in range(200):
        acts = {i: {'robot': rng.uniform(-1, 1, 7), 'human': rng.uniform(-1, 1, 10)} for i in range(n)}
        env.send_actions(acts)
        obs, rew, done, info, _ = env.poll()
        if k == 0:
            so, sr, sd, si = scalar.step(acts[5])
            assert np.abs(so['robot'] - obs[5]['robot']).max() < 1e-5 and np.abs(so['human'] - obs[5]['human']).max() < 1e-5 and abs(sr['robot'] - rew[5]['robot']) < 1e-5
        assert all(done[i]['__all__'] == (k == 199) for i in range(n)) and rew[3]['robot'] == rew[3]['human']
        assert set(info[0]['robot']) == {'total_force_on_human', 'task_success', 'action_robot_len', 'action_human_len', 'obs_robot_len', 'obs_human_len'}
    first = env.try_reset(3)
    assert first['robot'].shape == (30,) and np.isfinite(first['human']).all() and not np.array_equal(first['robot'], obs[3]['robot'])
    scalar.disconnect(); env.stop()


@pytest.mark.gpu
def test_multi_agent_batch_adapter_under_a_sampler_loop():
    """A stand-in for RLlib's sampler (ray 1.x _env_runner / _process_observations; ray itself is not installed): poll(), for every env_id with
    an observation read dones[env_id]['__all__'], the per-agent rewards and infos[env_id][agent]; an env whose '__all__' is set is reset with
    try_reset(env_id) and its first observation joins the round; actions for every observed agent go back through send_actions.  Runs over
    an episode boundary.  (The first poll of the adapter used to return empty reward / done / info dictionaries: KeyError in this loop.)"""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd.rllib import AgxMultiAgentBatchEnv
    n = 8
    env = AgxMultiAgentBatchEnv('assistive_gym:ScratchItchPR2Human-v1', n, pool_size=8)
    rng = np.random.RandomState(1)
    episodes, steps, ret = 0, 0, {i: 0.0 for i in range(n)}
    for rnd in range(203):
        obs, rewards, dones, infos, off = env.poll()
        to_eval = {}
        for env_id, agent_obs in obs.items():
            all_done = dones[env_id]['__all__']
            for agent, o in agent_obs.items():
                r = rewards[env_id][agent]
                assert (r is None) == (rnd == 0) and isinstance(infos[env_id][agent], dict)
                if r is not None and agent == 'robot':
                    ret[env_id] += r
            if all_done:
                episodes += 1
                first = env.try_reset(env_id)
                assert first is not None and set(first) == {'robot', 'human'} and np.isfinite(first['robot']).all()
                agent_obs = first
            to_eval[env_id] = agent_obs
        env.send_actions({i: {'robot': rng.uniform(-1, 1, 7), 'human': rng.uniform(-1, 1, 10)} for i in to_eval})
        steps += 1
    assert episodes == n and all(np.isfinite(v) and v < 0 for v in ret.values())
    env.stop()
