"""-m gpu: the HIP stepper (through the C ABI, include/agx.h) against the CPU oracle on the same
seeded inputs.  Tolerances: the north star asks for contact forces and rewards within 1e-3
relative of the reference; here the reference is the f64 oracle (PARITY UNPINNED vs PyBullet, see
oracle/agx_oracle.h) and the device computes in f32."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu_lib():
    from assistive_gym_amd import libagx
    L = libagx.load()
    if L.agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    return L


def test_wave_primitives(gpu_lib):
    from assistive_gym_amd import libagx
    libagx.check(gpu_lib.agx_selftest(0), 'agx_selftest')


def _settled_states(blob, n, seed):
    from assistive_gym_amd.vec_env import build_reset_pool
    return build_reset_pool(blob, n, seed)


def test_settle_matches_oracle(gpu_lib, blob, oracle):
    from assistive_gym_amd.host.reset import make_states
    from assistive_gym_amd.libagx import Stepper
    n = 16
    states, _ = make_states(blob, n, seed=4001)
    st = Stepper(blob, n)
    st.set_state(states)
    st.settle(3)
    st.synchronize()
    got = st.get_state()
    for i in range(n):
        ref = states[i].copy()
        oracle.settle(ref, 3)
        vg, vr = blob.view(got[i]), blob.view(ref)
        assert np.abs(vg['q'] - vr['q']).max() < 1e-5
        assert np.abs(vg['free'][0, :2, :7] - vr['free'][0, :2, :7]).max() < 1e-4      # tool, bowl pose
        assert np.abs(vg['free'][0, 2:, :3] - vr['free'][0, 2:, :3]).max() < 1e-4      # particle positions


def test_step_matches_oracle(gpu_lib, blob, oracle):
    """One env.step() from identical settled states with identical actions: obs, reward, forces."""
    from assistive_gym_amd.libagx import Stepper
    n, steps = 32, 6
    states = _settled_states(blob, n, 5001)
    st = Stepper(blob, n)
    rng = np.random.RandomState(7)
    ref_states = states.copy()
    worst = dict(obs=0.0, reward=0.0, force=0.0, q=0.0)
    flips = 0
    for k in range(steps):
        st.set_state(ref_states)            # re-synchronise every step: single-step parity
        actions = rng.uniform(-1, 1, (n, blob.act_dim)).astype(np.float32)
        obs, rew, done, info = st.step_host(actions)
        got = st.get_state()
        for i in range(n):
            o_obs, o_rew, o_done, o_info = oracle.step(ref_states[i], actions[i])
            worst['obs'] = max(worst['obs'], np.abs(obs[i] - o_obs).max())
            worst['reward'] = max(worst['reward'], abs(rew[i] - o_rew) / max(1.0, abs(o_rew)))
            worst['force'] = max(worst['force'], abs(info[i, 0] - o_info[0]) / max(1.0, abs(o_info[0])))
            worst['q'] = max(worst['q'], np.abs(blob.view(got[i])['q'] - blob.view(ref_states[i])['q']).max())
            assert bool(done[i]) == o_done
            # the 5th-substep contact set can differ by a borderline candidate (predicted gap within
            # rounding of the 1 mm slack after 4 substeps of f32 vs f64 drift); such a row carries no force
            flips += int(info[i, 6] != o_info[6])
    print('worst deviations', worst, 'contact-count flips', flips, 'of', n * steps)
    assert flips <= 0.03 * n * steps
    # measured on MI355X: obs / q deviations of ~1e-6 (f32 device vs f64 oracle); the bounds leave a 50x margin
    assert worst['obs'] < 1e-4 and worst['reward'] < 1e-4 and worst['force'] < 1e-3 and worst['q'] < 5e-5


def test_debug_internals_match_oracle(gpu_lib, blob, oracle):
    """First-substep contact set (order, distances) and M^-1 against the oracle."""
    import torch
    from assistive_gym_amd.libagx import Stepper, load
    n = 8
    states = _settled_states(blob, n, 6001)
    st = Stepper(blob, n)
    st.set_state(states)
    dev = torch.device('cuda', 0)
    act = torch.zeros((n, blob.act_dim), device=dev)
    obs = torch.zeros((n, blob.obs_dim), device=dev); rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, 8), device=dev)
    dw, o_con, o_minv, md = st.debug_layout()[:4]
    dbg = torch.zeros((n, dw), device=dev)
    st.step_dev(act, obs, rew, done, info, debug=dbg)
    torch.cuda.synchronize()
    dbg = dbg.cpu().numpy()
    for i in range(n):
        ref = states[i].copy()
        con = oracle.substep_debug(ref)
        nc = int(dbg[i, 0])
        assert nc == len(con)
        ce = dbg[i, o_con:o_con + 64 * 16].reshape(64, 16)[:nc]
        cei = ce.view(np.int32)
        assert np.array_equal(cei[:, 0], con[:, 0].astype(np.int32)) and np.array_equal(cei[:, 1], con[:, 1].astype(np.int32))
        assert np.abs(ce[:, 13] - con[:, 11]).max() < 1e-5
        Minv = dbg[i, o_minv:o_minv + md * md].reshape(md, md)[:blob.ndof, :blob.ndof]
        Mo = oracle.minv(states[i].copy())
        nr = blob.nrobot
        assert np.abs(Minv[:nr, :nr] - Mo[:nr, :nr]).max() / np.abs(Mo[:nr, :nr]).max() < 1e-4
        assert np.abs(Minv[nr:, nr:] - Mo[nr:, nr:]).max() <= 1e-4 * max(1.0, np.abs(Mo[nr:, nr:]).max())


def test_vec_env_rollout_properties(gpu_lib, blob):
    """Size-independent properties at the bench size: finite outputs, done exactly at step 200,
    auto-reset restores pool states, results independent of env placement (sharding)."""
    import torch
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    n = 4096
    env = FeedingJacoVecEnv(n, pool_size=64, seed=1001)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    first = None
    for k in range(3):
        a = torch.rand((n, blob.act_dim), device='cuda', generator=g) * 2 - 1
        obs, rew, done, info = env.step(a)
        if first is None:
            first = (obs.clone(), rew.clone(), a.clone())
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert int(done.sum()) == 0
    # placement independence: env i and env i+64 started from the same pool state; give them the
    # same action in a fresh run and they must agree bit for bit
    env.reset()
    a = first[2].clone(); a[64:128] = a[0:64]
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    assert torch.equal(obs[0:64], obs[64:128]) and torch.equal(rew[0:64], rew[64:128])
    env.close()


def test_full_episode_invariants_at_bench_size(gpu_lib, blob):
    """BASELINE config 2 at full size (4096 envs, a whole 200-step episode + auto-reset): properties that
    do not need the oracle -- finite outputs, done exactly at step 200, unit quaternions, robot joints
    inside their limits, the tool still attached to the end effector, particle bookkeeping, and bitwise
    run-to-run determinism."""
    import torch
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    from assistive_gym_amd.host.kin import RobotKin
    n, T = 4096, 200

    def rollout():
        env = FeedingJacoVecEnv(n, pool_size=64, seed=1001)
        env.reset()
        g = torch.Generator(device='cuda'); g.manual_seed(11)
        ok = torch.ones((), dtype=torch.bool, device='cuda')
        early_done = torch.zeros((), dtype=torch.int64, device='cuda')
        ret = torch.zeros(n, device='cuda')
        env.auto_reset = False
        for k in range(T):
            a = torch.rand((n, blob.act_dim), device='cuda', generator=g) * 2 - 1
            obs, rew, done, info = env.step(a)
            ok &= torch.isfinite(obs).all() & torch.isfinite(rew).all() & torch.isfinite(info).all()
            if k < T - 1:
                early_done += done.sum()
            ret += rew
        torch.cuda.synchronize()
        final = env.stepper.get_state()
        out = (bool(ok), int(early_done), done.clone(), obs.clone(), ret.clone(), info.clone(), final)
        # auto-reset: every finished env takes a pool record and starts a new episode
        env.auto_reset = True
        env.stepper.reset_done(env.pool, env.pool_size, env.done)
        obs2, rew2, done2, info2 = env.step(torch.zeros((n, blob.act_dim), device='cuda'))
        torch.cuda.synchronize()
        after = env.stepper.get_state()
        env.close()
        return out, (done2.clone(), after)

    (ok, early, done, obs, ret, info, final), (done2, after) = rollout()
    assert ok and early == 0 and bool(done.all())                      # feeding.py:37: done = iteration >= 200
    v = blob.view(final)
    assert int(v['iteration'].min()) == T and int(v['iteration'].max()) == T
    qn = np.linalg.norm(v['free'][:, :, 3:7], axis=2)
    alive = v['food_alive'][:, None] >> np.arange(blob.nfood)[None] & 1
    body_ok = np.ones_like(qn, dtype=bool); body_ok[:, blob.h['FOOD0']:blob.h['FOOD0'] + blob.nfood] = alive.astype(bool)
    assert np.abs(qn - 1.0)[body_ok].max() < 1e-3
    kin = RobotKin(blob)
    q = v['q'][:, :blob.nrobot].astype(np.float64)
    lim = [d for d in range(blob.nrobot) if blob.robot_i(d, 'HAS_LIMIT')]
    assert (q[:, lim] > kin.lower[lim] - 0.02).all() and (q[:, lim] < kin.upper[lim] + 0.02).all()
    # tool.py:46-47: the spoon is held by a fixed constraint of 500 N; sampled envs keep it within 1.5 cm
    tb = blob.h['TOOL_BODY']
    for i in range(0, n, 257):
        tp, _ = kin.tool_pose(v['base'][i, :3].astype(np.float64), v['base'][i, 3:7].astype(np.float64), v['q'][i].astype(np.float64))
        assert np.linalg.norm(tp - v['free'][i, tb, :3]) < 0.015
    # particle bookkeeping (feeding.py:50-83): eaten + still alive <= total, success counter = eaten particles
    assert (v['task_success'] >= 0).all() and (v['task_success'] + alive.sum(1) <= blob.nfood).all()
    assert np.isfinite(ret.cpu().numpy()).all() and float(info[:, 0].min()) >= 0.0
    # a new episode has started everywhere
    va = blob.view(after)
    assert int(va['iteration'].max()) == 1 and not bool(done2.any())
    # determinism: the same rollout again is bit-identical
    (ok_b, early_b, done_b, obs_b, ret_b, info_b, final_b), _ = rollout()
    assert torch.equal(obs, obs_b) and torch.equal(ret, ret_b) and np.array_equal(final, final_b)


def test_free_running_episode_tracks_oracle(gpu_lib, blob, oracle):
    """A whole 200-step episode WITHOUT re-synchronisation: the robot trajectory stays within a few
    milliradians of the oracle's, episode returns agree up to the particle events that chaos reorders
    (a spill is -5, feeding.py:50-83).  Measured on MI355X: median |dq| 9e-4 rad at step 200, return
    correlation 0.99."""
    from assistive_gym_amd.libagx import Stepper
    n, T = 16, 200
    states = _settled_states(blob, n, 7001)
    st = Stepper(blob, n)
    st.set_state(states)
    rng = np.random.RandomState(5)
    ref = states.copy()
    ret_g, ret_o = np.zeros(n), np.zeros(n)
    for k in range(T):
        a = rng.uniform(-1, 1, (n, blob.act_dim)).astype(np.float32)
        obs, rew, done, info = st.step_host(a)
        ret_g += rew
        for i in range(n):
            ret_o[i] += oracle.step(ref[i], a[i])[1]
        if k == 24:
            # 25 steps in: still essentially the same trajectory.  A single environment may already have gone through a
            # contact event that rounding decides differently (measured: 15 of 16 below 5e-5, one at 3e-3), hence quantiles
            dq = np.sort(np.abs(blob.view(st.get_state())['q'] - blob.view(ref)['q']).max(1))
            assert dq[-2] < 1e-3 and np.median(dq) < 1e-4 and dq[-1] < 2e-2
    dq = np.abs(blob.view(st.get_state())['q'][:, :blob.nrobot] - blob.view(ref)['q'][:, :blob.nrobot]).max(1)
    print('free-running drift median %.2e max %.2e' % (np.median(dq), dq.max()), 'returns corr %.4f' % np.corrcoef(ret_g, ret_o)[0, 1])
    assert np.median(dq) < 1e-2 and dq.max() < 0.1
    assert np.corrcoef(ret_g, ret_o)[0, 1] > 0.95
    assert abs(ret_g.mean() - ret_o.mean()) < 5.0            # at most one particle event per env on average


def test_scalar_env_facade(gpu_lib, blob, oracle):
    """FeedingJacoEnv: the reference's gym surface (feeding_envs.py:29-31) on top of a 1-env handle."""
    from assistive_gym_amd.envs import FeedingJacoEnv, make
    env = make('assistive_gym:FeedingJaco-v1')
    assert isinstance(env, FeedingJacoEnv)
    assert env.seed(1001) == [1001]
    obs = env.reset()
    assert obs.shape == (25,) and obs.dtype == np.float64 and np.isfinite(obs).all()
    state = env.get_state().copy()
    a = env.action_space.sample()
    obs, rew, done, info = env.step(a)
    o_obs, o_rew, o_done, o_info = oracle.step(state, a)
    assert np.abs(obs - o_obs).max() < 1e-4 and abs(rew - o_rew) < 1e-4 and done == o_done
    assert set(info) == {'total_force_on_human', 'task_success', 'action_robot_len', 'action_human_len', 'obs_robot_len', 'obs_human_len'}
    assert info['action_robot_len'] == 7 and info['obs_robot_len'] == 25
    for k in range(199):
        obs, rew, done, info = env.step(env.action_space.sample())
        assert done == (k == 198)
    with pytest.raises(ValueError):
        env.step(np.zeros(3))
    env.disconnect()


def test_food_events_match_oracle(gpu_lib, blob, oracle):
    """Finish kernel state machine (feeding.py:50-83) on the device: a particle just outside the 0.1 m
    spoon query is a spill (-5), one released above the mouth target is eaten (+20)."""
    from assistive_gym_amd.libagx import Stepper
    n = 8
    states = _settled_states(blob, n, 5101)
    food0 = blob.h['FOOD0']
    for i in range(n):
        v = blob.view(states[i])
        # even envs: diagonally away (near the edge of the 0.1 m query: spill or not depends on the spoon pose); odd envs: 0.25 m
        # straight above the spoon (it falls 5 cm during the step): no part of the spoon within 0.1 m
        v['free'][0, food0 + 0, :3] += np.array([-0.095, -0.095, 0.095] if i % 2 == 0 else [0.0, 0.0, 0.25], dtype=np.float32)
        v['free'][0, food0 + 0, 7:13] = 0.0
        v['free'][0, food0 + 1, :3] = v['target'][0] + np.array([0.0, 0.0, 0.045], dtype=np.float32)
        v['free'][0, food0 + 1, 7:13] = 0.0
    st = Stepper(blob, n)
    st.set_state(states)
    actions = np.zeros((n, blob.act_dim), dtype=np.float32)
    obs, rew, done, info = st.step_host(actions)
    got = st.get_state()
    eaten, spilled = 0, 0
    for i in range(n):
        ref = states[i].copy()
        o_obs, o_rew, o_done, o_info = oracle.step(ref, actions[i])
        vg, vr = blob.view(got[i]), blob.view(ref)
        assert int(vg['food_alive'][0]) == int(vr['food_alive'][0]) and int(vg['food_active'][0]) == int(vr['food_active'][0])
        assert int(vg['task_success'][0]) == int(vr['task_success'][0])
        assert abs(rew[i] - o_rew) < 1e-3 * max(1.0, abs(o_rew))
        eaten += int(vr['task_success'][0])
        spilled += 1 - (int(vr['food_alive'][0]) & 1)
    # the scenario exercises both events (spoon pose differs per env, so not necessarily in every env)
    assert eaten >= 1 and spilled >= 1


def test_coop_matches_oracle(gpu_lib, blob):
    """Co-op flavour (FeedingJacoHumanEnv): 11 actions, 48 observations, controllable head joints with
    per-environment limit scale and the tremor branch of take_step (env.py:201-215); single-step parity
    with the oracle over random impairments, then the dictionary surface of the scalar env."""
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    from oracle_lib import Oracle
    co = blob.coop()
    orc = Oracle(co)
    n, steps = 16, 4
    states = build_reset_pool(co, n, 5201)
    assert (co.view(states)['frozen'] == 0).all()
    st = Stepper(co, n)
    rng = np.random.RandomState(9)
    ref = states.copy()
    worst = dict(obs=0.0, reward=0.0, q=0.0)
    for k in range(steps):
        st.set_state(ref)
        a = rng.uniform(-1, 1, (n, co.act_dim)).astype(np.float32)
        obs, rew, done, info = st.step_host(a)
        got = st.get_state()
        for i in range(n):
            o_obs, o_rew, o_done, o_info = orc.step(ref[i], a[i])
            worst['obs'] = max(worst['obs'], np.abs(obs[i] - o_obs).max())
            worst['reward'] = max(worst['reward'], abs(rew[i] - o_rew) / max(1.0, abs(o_rew)))
            worst['q'] = max(worst['q'], np.abs(co.view(got[i])['q'] - co.view(ref[i])['q']).max())
            assert np.abs(co.view(got[i])['tremor_target'] - co.view(ref[i])['tremor_target']).max() < 1e-6
    print('co-op worst deviations', worst)
    assert obs.shape == (n, 48) and worst['obs'] < 1e-4 and worst['reward'] < 1e-4 and worst['q'] < 5e-5
    st.close()
    from assistive_gym_amd.envs import make
    env = make('assistive_gym:FeedingJacoHuman-v1')
    o = env.reset()
    assert set(o) == {'robot', 'human'} and o['robot'].shape == (25,) and o['human'].shape == (23,)
    o, r, d, info = env.step({'robot': env.action_space_robot.sample(), 'human': env.action_space_human.sample()})
    assert set(r) == {'robot', 'human'} and r['robot'] == r['human'] and set(d) == {'robot', 'human', '__all__'}
    assert info['robot']['action_human_len'] == 4 and info['human']['obs_human_len'] == 23
    env.disconnect()


@pytest.mark.parametrize('param,value,n', [('MAX_CONTACTS', 12, 3), ('MAX_ROWS', 60, 1), ('MAX_ENTRIES', 500, 5)])
def test_budget_truncation_and_ragged_batches(gpu_lib, blob, param, value, n):
    """Edge cases on the device: exhausted contact / row / coefficient budgets (same truncation as the
    oracle) on batches of 1, 3 and 5 environments (grids that are not a multiple of anything)."""
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.host.reset import make_states
    from oracle_lib import Oracle
    b = blob.set_param(param, value)
    orc = Oracle(b)
    states, _ = make_states(b, n, seed=5301)
    st = Stepper(b, n)
    st.set_state(states); st.settle(3); st.synchronize()
    ref = st.get_state()
    a = np.random.RandomState(3).uniform(-1, 1, (n, b.act_dim)).astype(np.float32)
    obs, rew, done, info = st.step_host(a)
    for i in range(n):
        o_obs, o_rew, o_done, o_info = orc.step(ref[i], a[i])
        assert info[i, 6] == o_info[6] and info[i, 7] == o_info[7]
        assert np.abs(obs[i] - o_obs).max() < 1e-4 and abs(rew[i] - o_rew) < 1e-3
    if param == 'MAX_CONTACTS':
        assert info[:, 6].max() <= value
    st.close()


def test_golden_trajectory_replay(gpu_lib, blob):
    """The committed oracle trajectory (tests/golden/feeding_jaco_oracle_traj.npz) replayed free-running through the C ABI."""
    import os
    from assistive_gym_amd.libagx import Stepper
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeding_jaco_oracle_traj.npz'))
    st = Stepper(blob, 1)
    st.set_state(g['state0'][None])
    for k in range(len(g['actions'])):
        obs, rew, done, info = st.step_host(g['actions'][k][None])
        assert np.abs(obs[0] - g['obs'][k]).max() < 2e-4 and abs(float(rew[0]) - float(g['reward'][k])) < 2e-4, k
    v, w = blob.view(st.get_state()), blob.view(g['state_end'][None].copy())
    assert np.abs(v['q'] - w['q']).max() < 1e-4 and np.abs(v['free'][0, :, :3] - w['free'][0, :, :3]).max() < 3e-3   # the particles jostle on the spoon: mm-level after 20 free-running steps
    assert v['food_alive'][0] == w['food_alive'][0] and v['iteration'][0] == w['iteration'][0]
    st.close()


@pytest.mark.parametrize('reset', ['pool', 'device'])
def test_nonfinite_environment_is_flagged_masked_and_replaced(gpu_lib, blob, reset):
    """one environment of the batch goes NaN mid-episode -> it alone reports done with a zero reward and the AGX_INFO_NONFINITE
    marker; it is replaced AT ONCE (reset='pool': from the pool; reset='device': from the rescue pool, joining the batch at its current
    iteration so that the masked re-sampling at the 200-step boundary finds the batch in lock step) and its row of the observations is
    the first observation of its new episode -- what a vector-env consumer expects with done -- not a stream of zeroed 1-step episodes"""
    import torch
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    n = 64
    env = FeedingJacoVecEnv(n, pool_size=16, seed=4242, reset=reset)
    env.reset()
    a = torch.zeros((n, env.act_dim), device='cuda')
    for _ in range(3):
        env.step(a)
    st = env.stepper.get_state()
    blob.view(st)['q'][5, 1] = np.nan
    blob.view(st)['free'][9, 0, 0] = np.inf
    env.stepper.set_state(st)
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    bad = (info[:, 6] >= 1.0e6).cpu().numpy()
    assert list(np.nonzero(bad)[0]) == [5, 9]
    assert done.cpu().numpy()[bad].all() and not done.cpu().numpy()[~bad].any()
    assert float(rew[5]) == 0.0 and float(rew[9]) == 0.0 and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    now = env.stepper.get_state()
    assert np.isfinite(now[:, :42]).all()
    fresh = env.stepper.observe_host()
    assert np.array_equal(obs.cpu().numpy()[bad], fresh[bad]) and np.abs(fresh[bad]).max() > 0     # first observation of the replacement
    it = blob.view(now)['iteration']
    if reset == 'device':
        assert (it == 4).all()                                # the replacements joined the batch at its iteration
    else:
        assert (it[~bad] == 4).all() and (it[bad] == 0).all()   # pool records start their own 200-step episode
    for _ in range(3):
        obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and not (info[:, 6] >= 1.0e6).any() and not done.any()
    env.close()


def test_oracle_parity_at_bench_size(gpu_lib, blob, oracle):
    """The BASELINE configuration itself (4096 environments in one handle: three chunks on internal streams) against the oracle: after 40 random-policy steps, one more step from the states as they
    are; 64 of the 4096 environments, spread over the chunks, are compared one by one"""
    import torch
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    n = 4096
    env = FeedingJacoVecEnv(n, pool_size=64, seed=2002)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(40):
        env.step(torch.rand((n, blob.act_dim), device='cuda', generator=g) * 2 - 1)
    torch.cuda.synchronize()
    before = env.stepper.get_state()
    a = torch.rand((n, blob.act_dim), device='cuda', generator=g) * 2 - 1
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    obs, rew, info, a = obs.cpu().numpy(), rew.cpu().numpy(), info.cpu().numpy(), a.cpu().numpy()
    worst = dict(obs=0.0, reward=0.0, force=0.0)
    flips = 0
    picks = [(i * 67) % n for i in range(64)]
    assert len({p % 4 for p in picks}) == 4 and min(picks) < 1365 < max(picks)
    for i in picks:
        s = before[i].copy()
        o_obs, o_rew, o_done, o_info = oracle.step(s, a[i])
        flips += int(info[i, 6] != o_info[6])             # a borderline (speculative) contact candidate decided differently in f32 and f64: compared like every other
        worst['obs'] = max(worst['obs'], float(np.abs(obs[i] - o_obs).max()))
        worst['reward'] = max(worst['reward'], abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)))
        worst['force'] = max(worst['force'], abs(float(info[i, 0]) - float(o_info[0])) / max(1.0, abs(float(o_info[0]))))
    print('bench-size parity:', worst, 'contact-count flips', flips, 'of 64 (compared too)')
    assert flips <= 4 and worst['obs'] < 1e-4 and worst['reward'] < 1e-4 and worst['force'] < 1e-3
    __import__('conditioning').tally('plain', 3 * len(picks))
    env.close()


@pytest.mark.parametrize('workload', ['feeding', 'wiping'])
def test_noop_retest_rule_against_the_plain_solve(gpu_lib, workload):
    """the no-op re-test rule (AGX_P_NOOP_RETEST = 5: a non-friction row whose visit at a re-test sweep changed nothing is skipped
    for the next 4 sweeps) is an APPROXIMATION that the oracle shares -- every other parity test is green by construction.  Here the device
    runs with the rule and the ORACLE WITHOUT it (NOOP_RETEST = 0: the plain 50-sweep solve): 64 environments x 20 steps of BASELINE config
    2 and of config 3's contact-rich workload (pad pressed onto the arm), each step from the device's own state; reward,
    total_force_on_human and the tool force must agree to 1e-3 relative (a step beyond it is judged against the plain oracle's 1-ulp
    sensitivity, tests/conditioning.py)."""
    import sys, os
    import torch
    import conditioning as C
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    from oracle_lib import Oracle
    n, steps = 64, 20
    if workload == 'feeding':
        b = ModelBlob.load('feeding_jaco')
        states = build_reset_pool(b, n, seed=6006)
        scale = 1.0
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b = ModelBlob.load('bed_bathing_sawyer')
        states = wiping_pool(b, n, 6006)
        b.view(states)['iteration'][:] = 0
        scale = 0.15
    assert b.param('NOOP_RETEST') == 5
    plain = Oracle(b.set_param('NOOP_RETEST', 0.0))
    st = Stepper(b, n)
    st.set_state(states)
    rng = np.random.RandomState(11)
    f = b.obs_dim_robot - 1                                  # the tool force entry of the observation
    worst = dict(reward=0.0, total_force=0.0, tool_force=0.0, obs=0.0)
    conditioned, touching = 0, 0
    for k in range(steps):
        ref = st.get_state()
        act = (rng.uniform(-1, 1, (n, b.act_dim)) * scale).astype(np.float32)
        obs, rew, done, info = st.step_host(act)
        for i in range(n):
            o_obs, o_rew, o_done, o_info = plain.step(ref[i].copy(), act[i])
            dev = dict(reward=abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)), total_force=abs(info[i, 0] - o_info[0]) / max(1.0, abs(o_info[0])),
                       tool_force=abs(obs[i, f] - o_obs[f]) / max(1.0, abs(o_obs[f])), obs=float(np.abs(np.delete(obs[i] - o_obs, f)).max()))
            touching += int(o_info[0] > 0 or o_obs[f] > 0)
            ff = C.force_floor(b)
            floor = dict(reward=0.06 * ff / max(1.0, abs(o_rew)), total_force=ff / max(1.0, abs(o_info[0])), tool_force=ff / max(1.0, abs(o_obs[f])), obs=0.0)
            cache = {}

            def level(eps, kk, _i=i, _ref=ref, _act=act, _cache=cache, _o=(o_obs, o_rew, o_info)):
                # the plain oracle's own response to a 1-ulp (eps None) / 1e-6 relative perturbation of its input, scaled like `dev`
                if eps not in _cache:
                    sn = C.ulp_sensitivity(b.set_param('NOOP_RETEST', 0.0), plain, _ref[_i], _act[_i], trials=4 if eps is None else 6, rel_eps=eps)
                    _cache[eps] = dict(reward=kk * sn['reward'] / max(1.0, abs(_o[1])), total_force=kk * sn['info'][0] / max(1.0, abs(_o[2][0])),
                                       tool_force=kk * sn['obs'][f] / max(1.0, abs(_o[0][f])), obs=kk * float(np.delete(sn['obs'], f).max()))
                return _cache[eps]
            for key in dev:
                ok, lim = C.check(dev[key], 1e-3, floor[key], ulp=lambda: level(None, C.K)[key], step=lambda: level(C.STEP_EPS, C.K_STEP)[key])
                assert ok, (workload, k, i, key, dev, lim, floor)
            if cache:
                conditioned += 1
                if C.STEP_EPS in cache:
                    print('step-level conditioning: %s step %d env %d dev %s' % (workload, k, i, {q: float('%.3g' % v) for q, v in dev.items()}))
                    os.makedirs('gpurun_out', exist_ok=True)
                    np.savez('gpurun_out/noop_parity_case_%s_%d_%d.npz' % (workload, k, i), start=ref[i], action=act[i], dev_obs=obs[i], dev_info=info[i], oracle_obs=o_obs, oracle_info=o_info)
            else:
                for key in dev:
                    worst[key] = max(worst[key], dev[key])
    st.close()
    print('%s: device (NOOP_RETEST 5) vs plain 50-sweep oracle, worst relative deviations %s; %d of %d steps judged by conditioning; %d steps with a force on the person / the tool'
          % (workload, {k: float('%.3g' % v) for k, v in worst.items()}, conditioned, n * steps, touching))
    assert conditioned <= 0.02 * n * steps
    assert workload == 'feeding' or touching > 0.1 * n * steps      # (free-running under random actions: 175 of 1280 in session r04h)


@pytest.mark.parametrize('workload', ['feeding', 'wiping'])
def test_warm_start_switch_on_the_device(gpu_lib, workload):
    """AGX_P_WARMSTART through the C ABI against the oracle's switch.  (a) 16 environments, three steps, every step from an injected state
    (agx_set_state forgets the memory; the oracle's is cleared per environment): substeps 2-5 of each step start warm on both sides.
    (b) ONE environment stepped four times without injection: the memory persists from step to step on both sides."""
    import sys, os
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    from oracle_lib import Oracle
    n = 16
    if workload == 'feeding':
        b0 = ModelBlob.load('feeding_jaco'); states = build_reset_pool(b0, n, seed=7007); scale = 1.0
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b0 = ModelBlob.load('bed_bathing_sawyer'); states = wiping_pool(b0, n, 7007); b0.view(states)['iteration'][:] = 0; scale = 0.15
    b = b0.set_param('WARMSTART', 0.85)
    o, cold = Oracle(b), Oracle(b0)
    st = Stepper(b, n)
    rng = np.random.RandomState(3)
    ref = states.copy()
    differs = 0.0
    for k in range(3):
        st.set_state(ref)
        act = (rng.uniform(-1, 1, (n, b.act_dim)) * scale).astype(np.float32)
        obs, rew, done, info = st.step_host(act)
        got = st.get_state()
        for i in range(n):
            o.forget_warm()
            sc = ref[i].copy()
            o_obs, o_rew, _, o_info = o.step(ref[i], act[i])
            cold.step(sc, act[i])
            differs = max(differs, float(np.abs(ref[i] - sc)[:b.h['S_ENV']].max()))
            fcol = b.obs_dim_robot - 1                                  # the tool-force entry: a force, bounded like info[0] below
            assert np.abs(np.delete(obs[i] - o_obs, fcol)).max() < 1e-4 and abs(rew[i] - o_rew) < 1e-4 * max(1.0, abs(o_rew)) + 0.06 * 1e-2, (workload, k, i)
            assert abs(obs[i, fcol] - o_obs[fcol]) <= max(1e-3 * max(1.0, abs(o_obs[fcol])), 1e-2), (workload, k, i)
            assert abs(info[i, 0] - o_info[0]) <= max(1e-3 * max(1.0, abs(o_info[0])), 1e-2), (workload, k, i, info[i, 0], o_info[0])
            assert np.abs(b.view(got[i:i + 1])['q'][0] - b.view(ref[i:i + 1])['q'][0]).max() < 1e-4
    assert differs > 1e-6 or workload == 'wiping'           # the switch changes the food pile's unconverged solve (the pad's small system converges either way)
    st.close()
    one = Stepper(b, 1)
    s = states[:1].copy()
    one.set_state(s); o.forget_warm()
    so = s[0].copy()
    for k in range(4):
        a = (rng.uniform(-1, 1, (1, b.act_dim)) * scale).astype(np.float32)
        obs, rew, done, info = one.step_host(a)
        o_obs, o_rew, _, o_info = o.step(so, a[0])
        assert np.abs(obs[0] - o_obs).max() < 3e-4 * (k + 1), (workload, 'persistent', k)      # free running: deviations compound
    o.forget_warm(); one.close()


@pytest.mark.gpu
@pytest.mark.parametrize('workload', ['feeding', 'wiping'])
def test_second_friction_direction_on_the_device(gpu_lib, workload):
    """AGX_P_FRICTION_DIRS = 2 through the C ABI against the oracle's switch: 16 environments, three steps, every step from an injected state.
    FeedingJaco takes the velocity-space sweep (the second block's row sets re-read per sweep, 47 contacts in the 160-row budget), the
    wiping workload of BedBathingSawyer the row-space sweep (or the velocity-space one where an environment's rows exceed its work area)."""
    import sys, os
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    from oracle_lib import Oracle
    import conditioning as C
    n = 16
    if workload == 'feeding':
        b0 = ModelBlob.load('feeding_jaco'); states = build_reset_pool(b0, n, seed=7107); scale = 1.0
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b0 = ModelBlob.load('bed_bathing_sawyer'); states = wiping_pool(b0, n, 7107); b0.view(states)['iteration'][:] = 0; scale = 0.15
    b = b0.set_param('FRICTION_DIRS', 2.0)
    o, one = Oracle(b), Oracle(b0)
    st = Stepper(b, n)
    rng = np.random.RandomState(4)
    ref = states.copy()
    differs, three, violent = 0.0, 0, 0
    fcol = b.obs_dim_robot - 1
    for k in range(3):
        st.set_state(ref)
        act = (rng.uniform(-1, 1, (n, b.act_dim)) * scale).astype(np.float32)
        obs, rew, done, info = st.step_host(act)
        got = st.get_state()
        for i in range(n):
            sc = ref[i].copy(); start = ref[i].copy()
            o_obs, o_rew, _, o_info = o.step(ref[i], act[i])
            c_info = one.step(sc, act[i])[3]
            differs = max(differs, float(np.abs(ref[i] - sc)[:b.h['S_ENV']].max()))
            assert info[i, 6] == o_info[6] and info[i, 7] == o_info[7], (workload, k, i, info[i], o_info)
            three += int(o_info[6] > 0 and o_info[7] - 3 * o_info[6] == c_info[7] - 2 * c_info[6])
            pose_tol = 1e-4
            dpose = float(np.abs(np.delete(obs[i] - o_obs, fcol)).max())
            if dpose >= pose_tol:
                # a VIOLENT step (tests/test_reference_pinned.py check_step_conditioned): the pad pressed with > 50 N onto the arm through eleven
                # redundant contacts with three rows each -- session r04h: 131 N, device 6.2e-4, the same kernel sources on the CPU wave emulator
                # 1.7e-4, every other environment of the batch 1e-7 ... 1e-6.  Judged against the oracle's own response to a 1e-5 relative
                # perturbation of its input (K x), as the spoon-on-face starts are.
                assert max(o_obs[fcol], o_info[0]) > 50.0, (workload, k, i, dpose, o_obs[fcol], o_info[0])
                sens = C.ulp_sensitivity(b, o, start, act[i], trials=8, rel_eps=1e-5)
                pose_tol = max(pose_tol, C.K * float(np.delete(sens['obs'], fcol).max()))
                violent += 1
                print('VIOLENT STEP %s step %d env %d: tool force %.0f N, pose deviation %.3g, bound %.3g' % (workload, k, i, o_obs[fcol], dpose, pose_tol))
            assert dpose < pose_tol and abs(rew[i] - o_rew) < pose_tol * max(1.0, abs(o_rew)) + 0.06 * max(1e-3 * max(1.0, abs(o_info[0])), C.force_floor(b)), (workload, k, i)
            assert abs(obs[i, fcol] - o_obs[fcol]) <= max(1e-3 * max(1.0, abs(o_obs[fcol])), C.force_floor(b)), (workload, k, i)
            assert abs(info[i, 0] - o_info[0]) <= max(1e-3 * max(1.0, abs(o_info[0])), C.force_floor(b)), (workload, k, i, info[i, 0], o_info[0])
            assert np.abs(b.view(got[i:i + 1])['q'][0] - b.view(ref[i:i + 1])['q'][0]).max() < pose_tol
    assert three > 0 and differs > 1e-7 and violent <= 2, (three, differs, violent)
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize('workload', ['feeding', 'wiping'])
def test_persistent_manifold_on_the_device(gpu_lib, workload):
    """AGX_P_MANIFOLD through the C ABI against the oracle's switch (the variant's second build kernel, agx_build_mf_kernel): ONE environment
    per handle stepped six times -- the cached points live in the environment's scratch record from step to step; the oracle's memory is
    process-wide, hence one environment at a time -- for six start states.  After every step the device continues from the ORACLE's state
    (written into the state tensor directly: agx_set_state would clear the memory), so every step is a single-step comparison with the
    memories of both sides built up over the steps before.  Contact and row counts must agree exactly."""
    import sys, os
    import torch
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    from oracle_lib import Oracle
    import conditioning as C
    n = 6
    if workload == 'feeding':
        b0 = ModelBlob.load('feeding_jaco'); states = build_reset_pool(b0, n, seed=7207); scale = 1.0
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b0 = ModelBlob.load('bed_bathing_sawyer'); states = wiping_pool(b0, n, 7207); b0.view(states)['iteration'][:] = 0; scale = 0.15
    b = b0.set_param('MANIFOLD', 1.0)
    o, plain = Oracle(b), Oracle(b0)
    fcol = b.obs_dim_robot - 1
    differs, more, flips, cached, compared = 0.0, 0, 0, 0, 0
    for i in range(n):
        one = Stepper(b, 1)
        one.set_state(states[i:i + 1]); o.forget_warm()
        so, sp = states[i].copy(), states[i].copy()
        rng = np.random.RandomState(40 + i)
        for k in range(6):
            a = (rng.uniform(-1, 1, (1, b.act_dim)) * scale).astype(np.float32)
            obs, rew, done, info = one.step_host(a)
            o_obs, o_rew, _, o_info = o.step(so, a[0])
            p_info = plain.step(sp, a[0])[3]
            if info[0, 6] != o_info[6] or info[0, 7] != o_info[7]:
                flips += 1                                   # a contact on a threshold (slack, break distance) in float32: rare
            else:
                compared += 1
                assert np.abs(np.delete(obs[0] - o_obs, fcol)).max() < (3e-4 if workload == 'feeding' else 1e-4), (workload, i, k, np.abs(obs[0] - o_obs).max())
                assert abs(obs[0, fcol] - o_obs[fcol]) <= max(1e-3 * max(1.0, abs(o_obs[fcol])), C.force_floor(b)), (workload, i, k)
            more += int(o_info[6] > p_info[6])
            cached += int(len(o.manifold_get()) > 0)
            differs = max(differs, float(np.abs(so - sp)[:b.h['S_ENV']].max()))
            sp[:] = so
            one.state_tensor()[0].copy_(torch.from_numpy(so))
            torch.cuda.synchronize()
        one.close()
    o.forget_warm()
    # (wiping: since round 5 every contact of the pad with the person inside the break distance is solved anyway -- group flag bit 6 --, the cached
    # points can coincide with them)
    assert (differs > 1e-7 or workload == 'wiping') and (more > 0 or workload == 'wiping') and flips <= 2, (differs, more, flips)
    # ... so for wiping the positive statement is: the oracle's cache held points while the steps were compared (the switch did something
    # on the side the device is held to), and nearly every step WAS compared with the manifold on
    assert cached >= n and compared >= 6 * n - 2, (workload, cached, compared)


@pytest.mark.gpu
def test_split_impulse_threshold_on_the_device(gpu_lib):
    """AGX_P_SPLIT_PEN through the C ABI against the oracle's switch: the wiping workload with a threshold of 0.2 mm (pressed contacts lose their
    positional term), 16 environments, three steps from injected states."""
    import sys, os
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from oracle_lib import Oracle
    import conditioning as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import wiping_pool
    n = 16
    b0 = ModelBlob.load('bed_bathing_sawyer'); states = wiping_pool(b0, n, 7307); b0.view(states)['iteration'][:] = 0
    b = b0.set_param('SPLIT_PEN', 0.0002)
    o, plain = Oracle(b), Oracle(b0)
    st = Stepper(b, n)
    rng = np.random.RandomState(6)
    ref = states.copy()
    fcol = b.obs_dim_robot - 1
    differs = 0.0
    for k in range(3):
        st.set_state(ref)
        act = (rng.uniform(-1, 1, (n, b.act_dim)) * 0.15).astype(np.float32)
        obs, rew, done, info = st.step_host(act)
        for i in range(n):
            sp = ref[i].copy()
            o_obs, o_rew, _, o_info = o.step(ref[i], act[i])
            plain.step(sp, act[i])
            differs = max(differs, float(np.abs(ref[i] - sp)[:b.h['S_ENV']].max()))
            assert info[i, 6] == o_info[6] and info[i, 7] == o_info[7], (k, i, info[i], o_info)
            assert np.abs(np.delete(obs[i] - o_obs, fcol)).max() < 1e-4 and abs(rew[i] - o_rew) < 1e-4 * max(1.0, abs(o_rew)) + 0.06 * max(1e-3 * max(1.0, abs(o_info[0])), C.force_floor(b)), (k, i)
            assert abs(obs[i, fcol] - o_obs[fcol]) <= max(1e-3 * max(1.0, abs(o_obs[fcol])), C.force_floor(b)) and abs(info[i, 0] - o_info[0]) <= max(1e-3 * max(1.0, abs(o_info[0])), C.force_floor(b)), (k, i)
    assert differs > 1e-6
    st.close()
