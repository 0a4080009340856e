"""On-device rollout plumbing (assistive_gym_amd/rollout.py; the reference's consumer is RLlib PPO, learn.py:9-37)."""
import math

import numpy as np
import pytest
import torch

from assistive_gym_amd.rollout import GaussianMLPPolicy, gae, collect


def test_policy_shapes_and_log_prob():
    torch.manual_seed(0)
    pi = GaussianMLPPolicy(25, 7)
    assert [m.out_features for m in pi.pi if hasattr(m, 'out_features')] == [100, 100, 14]      # learn.py:16 fcnet_hiddens
    obs = torch.randn(33, 25)
    a, logp, v = pi.act(obs)
    assert a.shape == (33, 7) and logp.shape == (33,) and v.shape == (33,)
    logp2, v2 = pi.log_prob(obs, a)
    torch.testing.assert_close(logp, logp2, rtol=1e-4, atol=1e-4)
    mean, log_std, _ = pi(obs)
    ref = torch.distributions.Normal(mean, log_std.exp()).log_prob(a).sum(-1)
    torch.testing.assert_close(logp2, ref, rtol=1e-4, atol=1e-4)


def test_gae_matches_the_textbook_recursion():
    rng = np.random.RandomState(0)
    T, N, gamma, lam = 12, 5, 0.99, 0.95
    r, v = rng.randn(T, N), rng.randn(T + 1, N)
    d = (rng.rand(T, N) < 0.2)
    adv, ret = gae(torch.tensor(r), torch.tensor(v), torch.tensor(d.astype(np.uint8)), gamma, lam)
    want = np.zeros((T, N))
    for n in range(N):
        for t in range(T):
            acc, disc = 0.0, 1.0
            for k in range(t, T):
                live = 0.0 if d[k, n] else 1.0
                acc += disc * (r[k, n] + gamma * v[k + 1, n] * live - v[k, n])
                if d[k, n]:
                    break
                disc *= gamma * lam
            want[t, n] = acc
    np.testing.assert_allclose(adv.numpy(), want, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ret.numpy(), want + v[:-1], rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_collect_on_device(blob):
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    n, T = 256, 6
    env = FeedingJacoVecEnv(n, pool_size=32, seed=5)
    env.reset()
    torch.manual_seed(1)
    pi = GaussianMLPPolicy(env.obs_dim, env.act_dim).to(env.device)
    g = torch.Generator(device='cuda'); g.manual_seed(2)
    buf = collect(env, pi, T, g)
    assert buf['obs'].shape == (T, n, 25) and buf['actions'].shape == (T, n, 7) and buf['values'].shape == (T + 1, n)
    assert all(t.is_cuda for t in buf.values())
    assert all(torch.isfinite(buf[k]).all() for k in ('obs', 'actions', 'logp', 'rewards', 'values'))
    assert int(buf['dones'].sum()) == 0
    # the rollout is what the env would have produced step by step: replay the recorded actions
    env2 = FeedingJacoVecEnv(n, pool_size=32, seed=5)
    env2.reset()
    for t in range(T):
        assert torch.equal(env2.obs, buf['obs'][t])
        _, rew, _, _ = env2.step(buf['actions'][t].contiguous())
        assert torch.equal(rew, buf['rewards'][t])
    adv, ret = gae(buf['rewards'], buf['values'], buf['dones'])
    assert torch.isfinite(adv).all() and adv.shape == (T, n)
    env.close(); env2.close()


@pytest.mark.gpu
def test_episode_boundary_returns_the_reset_observation(blob):
    """vector-env convention at done: obs = first observation of the new episode, terminal_obs = last of the old"""
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    n = 64
    for mode in ('pool', 'device'):
        env = FeedingJacoVecEnv(n, pool_size=16, seed=9, reset=mode)
        env.reset()
        a = torch.zeros((n, env.act_dim), device='cuda')
        for k in range(199):
            env.step(a)
        last = env.obs.clone()
        obs, rew, done, info = env.step(a)
        assert bool(done.all()) and env.terminal_obs is not None and not torch.equal(env.terminal_obs, obs)
        fresh = torch.zeros_like(obs)
        env.stepper.observe_dev(fresh)
        torch.cuda.synchronize()
        assert torch.equal(fresh, obs)
        assert int(blob.view(env.stepper.get_state())['iteration'].max()) == 0
        env.close()
