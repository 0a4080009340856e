"""DrinkingJaco on the device -- the `drinking` kernel variant through the C ABI against what the reference's own DrinkingJacoEnv.step()
returned (tests/golden/ref_steps.npz).  Written in round 3 WITHOUT a GPU at hand (the variant was checked on the CPU wave emulator:
test_reference_pinned.py::test_emulator_drinking_step_matches_the_reference); the file sorts last so that a failure here cannot hide the
rest of a `-x` run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cases():
    import torch
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    import test_reference_pinned as T
    return T, [T.case(n) for n in T.NAMES if n.startswith('drinking')]


@pytest.mark.parametrize('coop', [False, True])
def test_gpu_drinking_step_matches_the_reference(cases, coop):
    from assistive_gym_amd import libagx
    from refcases import variant_blob
    T, cs = cases
    cs = [c for c in cs if c['coop'] == coop]
    b = variant_blob('drinking_jaco', coop, '')
    st = libagx.Stepper(b, len(cs))
    st.set_state(np.stack([c['state'] for c in cs]))
    st.set_cloth(np.stack([c['cloth'] for c in cs]))
    obs, rew, done, info = st.step_host(np.stack([c['action'] for c in cs]))
    out, water = st.get_state(), st.get_cloth()
    for i, c in enumerate(cs):
        T.check_step(b, c, obs[i], rew[i], done[i], info[i], tol=1e-4, ftol=1e-3)
        T.check_state(b, c, out[i], tol=1e-4)
        T.check_water(c['name'], water[i], tol=5e-4)
    st.close()


def test_gpu_drinking_episode_is_finite_and_pours(cases):
    """4 environments x 90 steps of the tipping action: every output finite, no environment keeps all its water, the masks only lose bits"""
    from assistive_gym_amd import libagx
    from refcases import variant_blob
    T, cs = cases
    c = [c for c in cs if c['name'] == 'drinking_none_step0'][0]
    b = variant_blob('drinking_jaco', False, '')
    n = 4
    st = libagx.Stepper(b, n)
    st.set_state(np.stack([c['state']] * n)); st.set_cloth(np.stack([c['cloth']] * n))
    a = np.zeros((n, 7), np.float32); a[:, 4], a[:, 5], a[:, 6] = 0.5, 1.0, 1.0
    t = b.h['S_TASK']
    prev = st.get_state().view(np.uint32)[:, t:t + 4].copy()
    total = np.zeros(n)
    for k in range(90):
        obs, rew, done, info = st.step_host(a)
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
        cur = st.get_state().view(np.uint32)[:, t:t + 4]
        assert not (cur & ~prev).any()
        prev = cur.copy(); total += info[:, 4]
    alive = np.array([sum(bin(int(x)).count('1') for x in row[:2]) for row in prev])
    assert (alive < 20).all() and np.allclose(total, -(64 - alive))
    st.close()


# ---- round 4: Drinking as a product path -- env ids, the batched environment with a (state, water) pool, resets on the device, every robot ---------
def _oracle_step_from(b, o, s, w, a):
    s, w = s.copy(), w.copy()
    obs, rew, done, info = o.step_cloth(s, w, a)
    return obs, rew, done, info, s, w


def test_gpu_200_step_episode_against_the_oracle():
    """DrinkingJacoVecEnv (pool of device-sampled, device-settled (state, water) pairs), one whole 200-step episode of 4 environments with the
    tipping action from step 100 on: EVERY step is repeated by the CPU oracle from the device's own pre-step state and water, and observation,
    reward, done, task_success, total force and the water events must agree; the episode ends at step 200 for every environment, the pool
    replaces state AND water"""
    import torch
    from assistive_gym_amd.vec_env import DrinkingJacoVecEnv
    from oracle_lib import Oracle
    n = 4
    env = DrinkingJacoVecEnv(n, pool_size=4, seed=515)
    b = env.blob
    o = Oracle(b)
    obs0 = env.reset()
    assert obs0.shape == (n, 25) and torch.isfinite(obs0).all() and env.stepper.variant() == 'drinking'
    rng = np.random.RandomState(9)
    worst = dict(obs=0.0, rew=0.0, water=0.0)
    events = 0
    for k in range(200):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32) * (0.3 if k < 100 else 1.0)
        if k >= 100:
            a[:, 4], a[:, 5], a[:, 6] = 0.5, 1.0, 1.0
        s0, w0 = env.stepper.get_state(), env.stepper.get_cloth()
        obs, rew, done, info = env.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        ob, rw, dn, inf = (obs if k < 199 else env.terminal_obs).cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy()
        for i in range(n):
            o_obs, o_rew, o_done, o_info, s1, w1 = _oracle_step_from(b, o, s0[i], w0[i], a[i])
            # a particle within rounding of the 0.1 m spill / 0.03 m mouth thresholds may flip between f32 and f64: then the water reward of this
            # step differs by a whole event (+-1, +10); everything else is compared on steps whose events agree
            if inf[i, 4] != o_info[4]:
                events += 1
                continue
            worst['obs'] = max(worst['obs'], float(np.abs(ob[i, :24] - o_obs[:24]).max()))
            worst['rew'] = max(worst['rew'], abs(float(rw[i]) - o_rew))
            assert bool(dn[i]) == o_done == (k == 199) and inf[i, 1] == o_info[1], (k, i)
            assert abs(inf[i, 0] - o_info[0]) <= 1e-3 * max(1.0, abs(o_info[0])), (k, i, inf[i, 0], o_info[0])      # total_force_on_human
    assert worst['obs'] < 2e-4 and worst['rew'] < 1e-3, worst
    assert events <= 4, events
    # the auto-reset at the boundary: states AND water come from the pool again
    s2, w2 = env.stepper.get_state(), env.stepper.get_cloth()
    v = b.view(s2)
    assert (v['iteration'] == 0).all() and any(np.array_equal(w2[0], env.cloth_pool_host[j]) for j in range(4))
    assert (v['task'][:, 0] == -1).all()                                                      # every particle alive again
    env.close()


@pytest.mark.parametrize('robot,cls', [('jaco', 'DrinkingJacoVecEnv'), ('sawyer', 'DrinkingSawyerVecEnv'), ('stretch', 'DrinkingStretchVecEnv')])
def test_gpu_device_reset_matches_its_restatement(robot, cls):
    """agx_sample_reset on the GPU against the numpy restatement (IK restarts / base pose search with the two start goals / placement draws),
    the water grid above the sampled cup, then the 50-step settle: the water ends up in the cup"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from assistive_gym_amd import libagx
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.host.reset_drinking import water_x0
    b = ModelBlob.load('drinking_' + robot)
    n = 8
    st = libagx.Stepper(b, n)
    st.sample_reset(4100)
    st.synchronize()
    s, w = st.get_state(), st.get_cloth()
    o = ro.with_collision_check(b.words)
    for i in range(n):
        want, info = o.sample(4100 + i)
        fl = np.ones(b.state_words, bool)
        assert np.abs(s[i] - want)[np.isfinite(want)].max() < 2e-6 or np.array_equal(s[i].view(np.uint32), want.view(np.uint32)), (robot, i)
        cup = b.view(s[i:i + 1])['free'][0, 0, :3]
        assert np.abs(w[i, 0] - (water_x0(b) + cup)).max() < 1e-6 and not w[i, 1].any()         # drinking.py:163-167
    st.settle(50); st.synchronize()
    s, w = st.get_state(), st.get_cloth()
    assert np.isfinite(s[:, :b.h['S_ENV']]).all() and np.isfinite(w).all()
    cup = b.view(s)['free'][:, 0, :3]
    assert (np.linalg.norm(w[:, 0] - cup[:, None], axis=2).max(axis=1) < 0.12).all()             # nothing left the cup while it dropped in
    assert np.linalg.norm(w[:, 1], axis=2).max() < 0.3
    st.close()


@pytest.mark.parametrize('robot', ['Jaco', 'Panda', 'Sawyer', 'Baxter', 'PR2', 'Stretch'])
@pytest.mark.parametrize('coop', [False, True])
def test_gpu_every_drinking_env_id_steps_like_the_oracle(robot, coop):
    """the 12 ids of drinking_envs.py through the scalar env of the drop-in package: reset() on the device, three steps, each against the oracle"""
    from assistive_gym_amd.envs import ENV_IDS
    from oracle_lib import Oracle
    env = ENV_IDS['Drinking%s%s-v1' % (robot, 'Human' if coop else '')]()
    env.seed(77)
    obs = env.reset()
    o = Oracle(env.blob)
    rng = np.random.RandomState(3)
    for k in range(3):
        s0, w0 = env.get_state()
        a = rng.uniform(-1, 1, env.blob.act_dim).astype(np.float32)
        act = {'robot': a[:env.action_robot_len], 'human': a[env.action_robot_len:]} if coop else a
        ob, rew, done, info = env.step(act)
        o_obs, o_rew, o_done, o_info, _, _ = _oracle_step_from(env.blob, o, s0, w0, a)
        flat = np.concatenate([ob['robot'], ob['human']]) if coop else ob
        r = rew['robot'] if coop else rew
        assert np.abs(flat - o_obs).max() < 2e-4 and abs(r - o_rew) < 1e-3, (robot, coop, k)
    env.disconnect()


def test_gpu_fresh_drinking_resets_every_episode():
    """reset='device': the masked agx_reset at the 200-step boundary samples every environment anew, water included"""
    import torch
    from assistive_gym_amd.vec_env import DrinkingJacoVecEnv
    n = 16
    env = DrinkingJacoVecEnv(n, seed=99, reset='device')
    env.reset()
    w0, s0 = env.stepper.get_cloth(), env.stepper.get_state()
    a = torch.zeros((n, 7), device='cuda')
    for k in range(200):
        obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    assert bool(done.all()) and torch.isfinite(obs).all()
    w1, s1 = env.stepper.get_cloth(), env.stepper.get_state()
    v0, v1 = env.blob.view(s0), env.blob.view(s1)
    assert (v1['iteration'] == 0).all() and (np.abs(v1['target'] - v0['target']).max(axis=1) > 1e-5).all()      # new head angles for every environment
    cup = v1['free'][:, 0, :3]
    assert (np.linalg.norm(w1[:, 0] - cup[:, None], axis=2).max(axis=1) < 0.12).all() and (v1['task'][:, 0] == -1).all()
    env.close()
