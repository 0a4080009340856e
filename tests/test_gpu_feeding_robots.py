"""-m gpu: FeedingSawyer-v1 / FeedingBaxter-v1 on the HIP stepper (feeding_l kernel variant: 320 colliders) against the CPU oracle, from
pool states built the product way (host base pose search + the device's collision pass + 25 settle steps).  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['sawyer', 'baxter', 'pr2'])
def rb(request):
    from assistive_gym_amd import libagx
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    b = ModelBlob.load('feeding_' + request.param)
    return request.param, b, Oracle(b)


def test_step_matches_oracle(rb):
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    name, b, oracle = rb
    n, steps = 16, 4
    states = build_reset_pool(b, n, 5001)
    st = Stepper(b, n)
    assert st.variant() == 'feeding_l'
    rng = np.random.RandomState(7)
    ref = states.copy()
    worst = dict(obs=0.0, reward=0.0, force=0.0, q=0.0)
    flips = 0
    for k in range(steps):
        st.set_state(ref)                   # single-step parity
        actions = rng.uniform(-1, 1, (n, b.act_dim)).astype(np.float32)
        obs, rew, done, info = st.step_host(actions)
        got = st.get_state()
        for i in range(n):
            o_obs, o_rew, o_done, o_info = oracle.step(ref[i], actions[i])
            worst['obs'] = max(worst['obs'], np.abs(obs[i] - o_obs).max())
            worst['reward'] = max(worst['reward'], abs(rew[i] - o_rew) / max(1.0, abs(o_rew)))
            worst['force'] = max(worst['force'], abs(info[i, 0] - o_info[0]) / max(1.0, abs(o_info[0])))
            worst['q'] = max(worst['q'], np.abs(b.view(got[i])['q'] - b.view(ref[i])['q']).max())
            assert bool(done[i]) == o_done
            flips += int(info[i, 6] != o_info[6])
    st.close()
    print('worst deviations', worst, 'contact-count flips', flips, 'of', n * steps)
    assert flips <= 0.08 * n * steps
    assert worst['obs'] < 1e-4 and worst['reward'] < 1e-4 and worst['force'] < 1e-3 and worst['q'] < 5e-5


def test_vec_env_rollout_and_scalar_env(rb):
    import torch
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.envs import make
    name, b, oracle = rb
    n = 64
    env = getattr(vec_env, 'Feeding%sVecEnv' % {'pr2': 'PR2'}.get(name, name.capitalize()))(n, pool_size=8, seed=3)
    obs = env.reset()
    assert obs.shape == (n, 25)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, 7), device='cuda', generator=g) * 2 - 1)
        assert bool(done.all()) == (k == 199)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert env.stepper.overflow_count() < 0.03 * n * 200 * 5
    env.close()
    e = make('assistive_gym:Feeding%s-v1' % {'pr2': 'PR2'}.get(name, name.capitalize()))
    o = e.reset()
    assert o.shape == (25,)
    o, r, d, info = e.step(e.action_space.sample())
    assert np.isfinite(r) and not d
    e.disconnect()
