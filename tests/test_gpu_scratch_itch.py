"""-m gpu: ScratchItchPR2-v1 and its co-op flavour (BASELINE config 4's environment) on the HIP stepper (scratch_itch kernel variant,
through the C ABI) against the CPU oracle on the same seeded inputs; single-step comparisons from the device's own states where the
trajectory is chaotic (sustained contact, classifier roll-backs).  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

from test_scratch_itch import _states, scratching_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def si():
    from assistive_gym_amd import libagx
    from assistive_gym_amd.blob import ModelBlob
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    return ModelBlob.load('scratch_itch_pr2')


@pytest.fixture(scope='module')
def si_oracle(si):
    from oracle_lib import Oracle
    return Oracle(si)


def _check_step(blob, o, st, ref, act, worst):
    obs, rew, done, info = st.step_host(act)
    got = st.get_state()
    f = blob.obs_dim_robot - 1
    for i in range(len(ref)):
        o_obs, o_rew, o_done, o_info = o.step(ref[i], act[i])
        # same contacts; the row count may differ by limit rows whose gap sits on the 0.25 rad activation distance (inactive either way):
        # the four finger joints are held at 0.25 rad, exactly that far from their lower limit (pr2.py:17)
        assert info[i, 6] == o_info[6] and abs(info[i, 7] - o_info[7]) <= 4, (i, info[i], o_info)
        dev = np.abs(obs[i] - o_obs)
        forces = [f] + ([blob.obs_dim - 2, blob.obs_dim - 1] if blob.is_coop else [])
        for k in forces:
            assert dev[k] <= 1e-3 * max(1.0, abs(o_obs[k])), (i, k, obs[i, k], o_obs[k])
            dev[k] = 0
        worst[i] = max(worst[i], float(dev.max()), abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)))
        assert info[i, 4] == o_info[4] and info[i, 1] == o_info[1] and bool(done[i]) == o_done
        for c in (0, 2, 3):
            assert abs(info[i, c] - o_info[c]) <= 1e-3 * max(1.0, abs(o_info[c])), (i, c, info[i], o_info)
        vg, vo = blob.view(got[i].reshape(1, -1)), blob.view(ref[i].reshape(1, -1))
        assert vg['task_success'][0] == vo['task_success'][0]
    return got


def test_variant(si):
    from assistive_gym_amd.libagx import Stepper
    st = Stepper(si, 2)
    assert st.variant() == 'scratch_itch' and st.debug_layout()[3] == 24
    st.close()


def test_step_matches_oracle(si, si_oracle):
    from assistive_gym_amd.libagx import Stepper
    a, _ = _states(si, 8, 6001)
    t, _ = _states(si, 4, 6101, impairment='tremor')
    w = [scratching_state(si, si_oracle, seed=6201 + k, depth=0.002 + 0.001 * k) for k in range(4)]
    states = np.concatenate([a, t, np.array(w)])
    n = len(states)
    st = Stepper(si, n)
    st.set_state(states)
    worst = np.zeros(n)
    scratches = 0
    ref = states.copy()
    for k in range(4):
        act = np.random.RandomState(100 + k).uniform(-1, 1, (n, 7)).astype(np.float32)
        act[12:] *= 0.1
        ref = st.get_state()                                   # single-step comparison from the device's own state
        got = _check_step(si, si_oracle, st, ref, act, worst)
        scratches = int(si.view(got)['task_success'][12:].sum())
    st.close()
    assert worst[:12].max() < 1e-4 and worst[12:].max() < 1e-3, worst
    assert scratches >= 3


def test_coop_matches_oracle(si):
    from assistive_gym_amd.libagx import Stepper
    from oracle_lib import Oracle
    coop = si.coop()
    o = Oracle(coop)
    states, _ = _states(coop, 8, 6301)
    st = Stepper(coop, 8)
    st.set_state(states)
    worst = np.zeros(8)
    for k in range(4):
        act = np.random.RandomState(200 + k).uniform(-1, 1, (8, 17)).astype(np.float32)
        ref = st.get_state()
        _check_step(coop, o, st, ref, act, worst)
    st.close()
    assert worst.max() < 2e-4, worst


def test_vec_env_rollout_and_scalar_env(si):
    import torch
    from assistive_gym_amd.envs import make
    from assistive_gym_amd.vec_env import ScratchItchPR2HumanVecEnv
    n = 64
    env = ScratchItchPR2HumanVecEnv(n, pool_size=8, seed=3)
    obs = env.reset()
    assert obs.shape == (n, 64) and env.act_dim == 17
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, 17), device='cuda', generator=g) * 2 - 1)
        assert bool(done.all()) == (k == 199)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and env.stepper.overflow_count() == 0
    env.close()
    e = make('assistive_gym:ScratchItchPR2Human-v1')
    o = e.reset()
    assert o['robot'].shape == (30,) and o['human'].shape == (34,)
    o, r, d, info = e.step({'robot': e.action_space_robot.sample(), 'human': e.action_space_human.sample()})
    assert np.isfinite(r['robot']) and not d['__all__'] and info['robot']['obs_human_len'] == 34
    e.disconnect()
    e = make('ScratchItchPR2-v1')
    o = e.reset()
    assert o.shape == (30,)
    e.disconnect()


def test_pool_refresher_swaps_fresh_states_in():
    """vec_env.PoolRefresher: a child process samples new start states (host sampler + device collision pass), the rollout swaps them into
    the pool at an episode boundary -- the fixed 256-state pool of the free-standing robots no longer repeats forever (VERDICT r2 item 5's
    interim measure).  sync mode: the boundary waits for the batch, so the test is deterministic."""
    import torch
    from assistive_gym_amd import libagx
    from assistive_gym_amd.vec_env import ScratchItchPR2VecEnv
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    env = ScratchItchPR2VecEnv(16, pool_size=8, seed=4242, pool_refresh=4, pool_refresh_sync=True)
    env.episode_len = 3                                             # a short "episode": the boundary logic is what is under test
    obs = env.reset()
    before = env.pool_host.copy()
    a = torch.zeros(16, env.act_dim, device='cuda')
    for _ in range(3):
        obs, rew, done, info = env.step(a)
    assert env.pool_refreshed == 4 and env._refresh_cursor == 4
    assert not np.array_equal(env.pool_host[:4], before[:4]) and np.array_equal(env.pool_host[4:], before[4:])
    assert np.isfinite(env.pool_host).all() and np.array_equal(env.pool.cpu().numpy(), env.pool_host)
    for _ in range(3):
        obs, rew, done, info = env.step(a)
    assert env.pool_refreshed == 8 and torch.isfinite(obs).all()
    env.close()
