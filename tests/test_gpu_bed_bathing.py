"""-m gpu: BedBathingSawyer-v1 (BASELINE config 3) on the HIP stepper (bed_bathing kernel variant, through the C ABI) against the
CPU oracle on the same seeded inputs.  Tolerances as in test_gpu_parity.py: forces / rewards within 1e-3 relative (north
star), observations far tighter; the reference here is the f64 oracle (PARITY UNPINNED vs PyBullet)."""
import numpy as np
import pytest

from bed_util import pad_pose
from test_bed_bathing import wiping_state, _states

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bed():
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd import libagx
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    return ModelBlob.load('bed_bathing_sawyer')


@pytest.fixture(scope='module')
def bed_oracle(bed):
    from oracle_lib import Oracle
    return Oracle(bed)


def _batch(bed, bed_oracle):
    """16 start states: sampled resets (all impairments), tremor-only ones, and crafted wiping contacts on both arm segments"""
    a, _ = _states(bed, 8, 5001)
    t, _ = _states(bed, 4, 5101, impairment='tremor')
    w = [wiping_state(bed, bed_oracle, seed=5201 + k, depth=0.002 + 0.002 * k, along=0.3 + 0.1 * k, arm='fore' if k % 2 == 0 else 'upper') for k in range(4)]
    return np.concatenate([a, t, np.array(w)])


def test_variant_and_limits(bed):
    from assistive_gym_amd.libagx import Stepper
    st = Stepper(bed, 4)
    assert st.variant() == 'bed_bathing'
    lay = st.debug_layout()
    assert lay[3] == 20 and lay[0] > lay[7]
    st.close()


def test_step_matches_oracle(bed, bed_oracle):
    from assistive_gym_amd.libagx import Stepper
    states = _batch(bed, bed_oracle)
    n = len(states)
    st = Stepper(bed, n)
    st.set_state(states)
    ref = states.copy()
    worst = np.zeros((n, 3))                                    # per env: obs, reward (relative), joint angles
    wiped = 0
    for k in range(4):
        act = np.random.RandomState(100 + k).uniform(-1, 1, (n, 7)).astype(np.float32)
        act[12:] *= 0.2                                         # the crafted contacts: stay on the arm for a few steps
        obs, rew, done, info = st.step_host(act)
        got = st.get_state()
        for i in range(n):
            o_obs, o_rew, o_done, o_info = bed_oracle.step(ref[i], act[i])
            assert info[i, 6] == o_info[6] and info[i, 7] == o_info[7], (k, i, info[i], o_info)
            worst[i, 0] = max(worst[i, 0], float(np.abs(obs[i] - o_obs).max()))
            worst[i, 1] = max(worst[i, 1], abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)))
            assert info[i, 4] == o_info[4] and info[i, 1] == o_info[1] and bool(done[i]) == o_done
            for c in (0, 2, 3):
                assert abs(info[i, c] - o_info[c]) <= 1e-3 * max(1.0, abs(o_info[c])), (k, i, c, info[i], o_info)
            vg, vo = bed.view(got[i].reshape(1, -1)), bed.view(ref[i].reshape(1, -1))
            worst[i, 2] = max(worst[i, 2], float(np.abs(vg['q'] - vo['q']).max()))
            assert np.array_equal(vg['task'], vo['task']) and vg['task_success'][0] == vo['task_success'][0]
            wiped += int(o_info[4])
    st.close()
    # free motion / tremor: f32 rounding only; the crafted wiping contacts (sustained tool-skin contact with friction, free running
    # for 4 steps on both sides): within the north star's 1e-3
    assert worst[:12].max() < 1e-4, worst[:12].max(0)
    assert worst[12:].max() < 1e-3, worst[12:].max(0)
    assert wiped >= 4                                           # the crafted states did wipe targets


def test_debug_internals_match_oracle(bed, bed_oracle):
    import torch
    from assistive_gym_amd.libagx import Stepper
    states = _batch(bed, bed_oracle)[8:]
    n = len(states)
    st = Stepper(bed, n)
    st.set_state(states)
    dev = torch.device('cuda', 0)
    act = torch.zeros((n, bed.act_dim), device=dev)
    obs = torch.zeros((n, bed.obs_dim), device=dev); rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, 8), device=dev)
    dw, o_con, o_minv, md = st.debug_layout()[:4]
    dbg = torch.zeros((n, dw), device=dev)
    st.step_dev(act, obs, rew, done, info, debug=dbg)
    torch.cuda.synchronize()
    dbg = dbg.cpu().numpy()
    for i in range(n):
        con = bed_oracle.substep_debug(states[i].copy())
        nc = int(dbg[i, 0])
        assert nc == len(con)
        ce = dbg[i, o_con:o_con + 64 * 16].reshape(64, 16)[:nc]
        cei = ce.view(np.int32)
        assert np.array_equal(cei[:, 0], con[:, 0].astype(np.int32)) and np.array_equal(cei[:, 1], con[:, 1].astype(np.int32))
        if nc:
            assert np.abs(ce[:, 13] - con[:, 11]).max() < 1e-5
        Minv = dbg[i, o_minv:o_minv + md * md].reshape(md, md)[:bed.ndof, :bed.ndof]
        Mo = bed_oracle.minv(states[i].copy())
        nr = bed.nrobot
        assert np.abs(Minv[:nr, :nr] - Mo[:nr, :nr]).max() / np.abs(Mo[:nr, :nr]).max() < 1e-4
        assert np.abs(Minv[nr:, nr:] - Mo[nr:, nr:]).max() <= 1e-4 * max(1.0, np.abs(Mo[nr:, nr:]).max())
        assert np.abs(Minv[:nr, nr:]).max() == 0 and np.abs(Minv[nr:, :nr]).max() == 0       # block diagonal
    st.close()


def test_free_running_episode_stays_close(bed, bed_oracle):
    """30 steps free running on both sides (no re-injection)"""
    from assistive_gym_amd.libagx import Stepper
    states, _ = _states(bed, 4, 5301)
    st = Stepper(bed, 4)
    st.set_state(states)
    import conditioning as C
    ref = states.copy()
    # a second oracle run from the same states moved by ONE float32 ulp per word: how far 30 free-running steps carry a perturbation the
    # device cannot even represent (tests/conditioning.py) -- the yardstick for the device's own drift
    twin = states.copy()
    fw = C.float_words(bed)
    twin[:, fw] = C._perturb_f32(twin[:, fw], np.random.RandomState(1))
    dev_max, spread = 0.0, 0.0
    for k in range(30):
        act = np.random.RandomState(300 + k).uniform(-1, 1, (4, 7)).astype(np.float32)
        obs, rew, done, info = st.step_host(act)
        for i in range(4):
            o_obs, o_rew, _, _ = bed_oracle.step(ref[i], act[i])
            t_obs, t_rew, _, _ = bed_oracle.step(twin[i], act[i])
            dev_max = max(dev_max, float(np.abs(obs[i] - o_obs).max()), abs(float(rew[i]) - o_rew))
            spread = max(spread, float(np.abs(t_obs - o_obs).max()), abs(t_rew - o_rew))
    st.close()
    print('free-running 30 steps: device vs oracle %.3g, oracle vs its 1-ulp twin %.3g' % (dev_max, spread))
    assert dev_max < max(2e-3, C.K * spread), (dev_max, spread)


def test_vec_env_rollout_and_auto_reset(bed):
    import torch
    from assistive_gym_amd.vec_env import BedBathingSawyerVecEnv
    n = 96
    env = BedBathingSawyerVecEnv(n, pool_size=8, seed=77)
    obs0 = env.reset().clone()
    assert obs0.shape == (n, 24) and torch.isfinite(obs0).all()
    # envs i and i + 8 start from the same pool entry
    assert torch.equal(obs0[0], obs0[8])
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    rsum = torch.zeros(n, device='cuda')
    for k in range(200):
        a = torch.rand((n, 7), device='cuda', generator=g) * 2 - 1
        obs, rew, done, info = env.step(a)
        rsum += rew
        assert bool(done.all()) == (k == 199)                   # done = iteration >= 200 (bed_bathing.py:31)
    assert torch.isfinite(rsum).all() and torch.isfinite(obs).all()
    assert env.stepper.overflow_count() == 0
    st = env.stepper.get_state()
    assert (bed.view(st)['iteration'] == 0).all()               # auto-reset from the pool
    env.close()


def test_scalar_env_surface(bed):
    from assistive_gym_amd.envs import BedBathingSawyerEnv, make
    env = make('assistive_gym:BedBathingSawyer-v1')
    assert isinstance(env, BedBathingSawyerEnv)
    env.seed(3)
    obs = env.reset()
    assert obs.shape == (24,) and obs.dtype == np.float64
    tot = 0.0
    for k in range(3):
        obs, r, d, info = env.step(env.action_space.sample())
        tot += r
        assert set(info) == {'total_force_on_human', 'task_success', 'action_robot_len', 'action_human_len', 'obs_robot_len', 'obs_human_len'}
    assert np.isfinite(tot) and not d and info['obs_robot_len'] == 24
    p, _ = pad_pose(bed, env.get_state())
    assert np.isfinite(p).all()
    env.disconnect()


def test_coop_with_arm_limit_classifier_matches_oracle(bed):
    """BedBathingSawyerHumanEnv: 17 actions, 52 observations, the arm-limit MLP after every substep (human.py:134-152)"""
    from assistive_gym_amd.libagx import Stepper
    from oracle_lib import Oracle
    import conditioning as C
    coop = bed.coop()
    o = Oracle(coop)
    states, _ = _states(coop, 8, 5401)
    # two envs start in an arm pose the classifier rejects, with a remembered valid pose: they are rolled back in the first substep
    from test_bed_bathing import _invalid_arm_pose
    v = coop.view(states)
    nr = coop.nrobot
    for i in (6, 7):
        v['task'][i, 6:10] = v['q'][i, nr + 3:nr + 7].view(np.int32); v['task'][i, 10] = 1
        bad = _invalid_arm_pose(coop, np.random.RandomState(i)).astype(np.float32)
        v['q'][i, nr + 3:nr + 7] = bad          # the motor targets keep the valid pose: after the roll-back the arm stays clear of the decision boundary
    st = Stepper(coop, 8)
    st.set_state(states)
    worst, rolled_back, where = 0.0, 0, None
    for k in range(5):
        # single-step comparisons from the device's own states: the co-op arm with the classifier in the loop is chaotic enough
        # (roll-backs are discontinuous) that free-running copies drift apart within a few steps
        ref = st.get_state()
        act = np.random.RandomState(400 + k).uniform(-1, 1, (8, 17)).astype(np.float32)
        act[6:, 7:] = 0
        obs, rew, done, info = st.step_host(act)
        got = st.get_state()
        for i in range(8):
            before = coop.view(ref[i].reshape(1, -1))['q'][0, nr + 3:nr + 7].copy()
            o_obs, o_rew, _, o_info = o.step(ref[i], act[i])
            dev = np.abs(obs[i] - o_obs)
            for f in (23, 50, 51):                              # contact forces (tool_force; total / pad force of the human's part): 1e-3 relative, or the float32 force floor (conditioning.force_floor)
                assert dev[f] <= max(1e-3 * max(1.0, abs(o_obs[f])), C.force_floor(coop)), (k, i, f, obs[i, f], o_obs[f])
                dev[f] = 0
            # the reward carries the force terms with weights <= 0.05 (env.py:249-256): their 1e-3 relative bound is part of its own
            fscale = max(1.0, float(np.abs(o_obs[[23, 50, 51]]).max()))
            d = max(float(dev.max()), max(0.0, abs(float(rew[i]) - o_rew) - 0.06 * 1e-3 * fscale) / max(1.0, abs(o_rew)))
            if d > worst:
                worst, where = d, (k, i, int(dev.argmax()), float(rew[i]), o_rew)
            vg, vo = coop.view(got[i].reshape(1, -1)), coop.view(ref[i].reshape(1, -1))
            assert vg['task'][0, 10] == vo['task'][0, 10] == 1
            assert np.abs(vg['task'][0, 6:10].view(np.float32) - vo['task'][0, 6:10].view(np.float32)).max() < 1e-5
            if k == 0 and i in (6, 7):                          # the rejected start pose was rolled back on both sides
                assert np.abs(vg['q'][0, nr + 3:nr + 7] - before).max() > 0.05
                rolled_back += 1
    st.close()
    assert rolled_back == 2
    # (8 environments x 5 steps with the classifier's roll-backs in the loop: the worst single entry of round 4's states was 3.6e-4, a joint angle
    # of the human's arm one step after a roll-back)
    assert obs.shape == (8, 52) and worst < 5e-4, (worst, where)


def test_coop_scalar_env_dicts(bed):
    from assistive_gym_amd.envs import make
    env = make('assistive_gym:BedBathingSawyerHuman-v1')
    obs = env.reset()
    assert set(obs) == {'robot', 'human'} and obs['robot'].shape == (24,) and obs['human'].shape == (28,)
    a = {'robot': env.action_space_robot.sample(), 'human': env.action_space_human.sample()}
    obs, rew, done, info = env.step(a)
    assert set(done) == {'robot', 'human', '__all__'} and rew['robot'] == rew['human'] and info['human']['action_human_len'] == 10
    env.disconnect()


# ---------------------------------------------------------------------------------------------------------------------
# the rag-doll settle of BedBathingEnv.reset (bed_bathing.py:119-137) on the device: bed_settle kernel variant
def _posed_batch(n, seed):
    from assistive_gym_amd.blob import ModelBlob
    from test_bed_settle import posed
    sb, bb = ModelBlob.load('bed_settle'), ModelBlob.load('bed_bathing_sawyer')
    rows, pres = zip(*[posed(sb, bb, seed + i) for i in range(n)])
    return sb, np.array(rows), pres


def test_ragdoll_settle_matches_oracle(bed):
    from assistive_gym_amd.libagx import Stepper
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.model import compiler as L
    sb_full, states, pres = _posed_batch(8, 7001)
    # With the bed's friction of 5 (bed_bathing.py:116; mu = 2.5 against skin) the friction bounds -+mu*lambda_n make the 50 projected
    # Gauss-Seidel sweeps non-contractive once the body lies on the mattress: oracle and device BOTH wander with the sweep count
    # (measured: 50 -> 200 -> 800 sweeps move the f64 result by 0.05 .. 0.1 rad/s) and rounding differences are amplified.  The
    # step-by-step comparison therefore runs on the same scene with bed friction 1 (contractive: identical results at 50 / 800 sweeps);
    # the reference's friction is used for the resting-place check below.
    w = sb_full.words.copy()
    for c in range(*sb_full.meta['ranges']['bed']):
        w.view(np.float32)[sb_full.h['OFF_COLL'] + c * L.C['STRIDE'] + L.C['FRICTION']] = 1.0
    sb = ModelBlob(w, sb_full.meta)
    o = Oracle(sb)
    st = Stepper(sb, 8)
    assert st.variant() == 'bed_settle'
    # single simulation steps from the f64 oracle's own trajectory -- in the air with the arms pushed out of the torso, at the
    # impact (around step 20), through the settle: compared step by step the f32 / f64 trajectories cannot drift apart chaotically
    ref = states.copy()
    worst = []
    for k, advance in enumerate((0, 6, 6, 6, 2, 2, 2, 6, 10, 20, 30)):
        for i in range(8):
            o.settle(ref[i], advance)
        st.set_state(ref)
        st.settle(1)
        got = st.get_state()
        nxt = ref.copy()
        for i in range(8):
            o.settle(nxt[i], 1)
        dq, dqd = np.abs(got[:, :47] - nxt[:, :47]).max(), np.abs(got[:, 47:94] - nxt[:, 47:94]).max()
        worst.append((float(dq), float(dqd)))
        assert dq < 5e-5 and dqd < 2.5e-3, (k, worst)          # dq = dt * dqd
    print('ragdoll single-step worst (dq, dqd):', max(w[0] for w in worst), max(w[1] for w in worst))
    # the whole 100-step settle from the start with the reference's friction: same resting place, loosely (the trajectories are chaotic)
    assert st.overflow_count() == 0
    st.close()
    o = Oracle(sb_full)
    st = Stepper(sb_full, 8)
    st.set_state(states)
    st.settle(100)
    got = st.get_state()
    ref = states.copy()
    for i in range(8):
        o.settle(ref[i], 100)
    assert np.abs(got[:, :3] - ref[:, :3]).max() < 0.02 and np.abs(got[:, 3:6] - ref[:, 3:6]).max() < 0.1
    assert st.overflow_count() == 0
    st.close()


def test_ragdoll_settler_rests_the_human_on_the_bed(bed):
    from assistive_gym_amd.host import reset_bed as rb
    from oracle_lib import Oracle
    sb, states, pres = _posed_batch(6, 7101)
    settler = rb.RagdollSettler(6)
    out = settler(states)
    o = Oracle(sb)
    bed0 = sb.meta['ranges']['bed'][0]
    for i in range(6):
        v = sb.view(out[i:i + 1])
        assert np.isfinite(out[i]).all()
        assert np.abs(v['qd'][0]).max() < 3.0 and np.abs(v['qd'][0, :3]).max() < 0.08
        bp, bq, hq = rb.settled_pose(sb, out[i:i + 1], pres[i]['hm'])
        assert 0.8 < bp[2] < 0.92 and abs(bp[0] + 0.15) < 0.1 and abs(bp[1] - 0.2) < 0.1
        assert np.max(np.maximum(pres[i]['hm'].lower - hq, hq - pres[i]['hm'].upper)) < 1e-2
        con = o.collide(out[i])                                  # the oracle's narrow phase on the DEVICE's resting state
        on_bed = con[con[:, 1] >= bed0]
        assert len(on_bed) >= 8 and on_bed[:, 11].min() > -8e-3


def test_bed_vec_env_pool_is_settled(bed):
    """BedBathingSawyerVecEnv's reset pool goes through the device settle; the stepper then runs from it"""
    import torch
    from assistive_gym_amd.vec_env import BedBathingSawyerVecEnv
    env = BedBathingSawyerVecEnv(8, pool_size=8, seed=4242)
    obs = env.reset()
    v = bed.view(env.pool_host)
    k = bed.meta['human_bodies'].index(-1)
    assert (v['human'][:, k, 2] > 0.8).all() and (v['human'][:, k, 2] < 0.92).all()      # the chest a chest radius above the mattress
    for _ in range(3):
        obs, rew, done, info = env.step(torch.zeros(8, 7, device=obs.device))
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()


def test_gjk_flat_tetrahedron_regression(bed):
    """tests/golden/settle_forearm_on_mattress_edge.npy (bed friction 1, captured by tools/gpu_settle_diag.py): the forearm lies along
    the mattress edge, 4 cm clear of that hull's core.  The device's GJK used to report an enclosed origin for the flat tetrahedron of
    that pair (side tests decided by the rounding of contracted multiply-adds; the CPU emulator did not reproduce it) and the pair came
    out 3.8 cm deep with normal +z.  An enclosure is now rejected while a separating plane is known."""
    import os
    import torch
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.model import compiler as L
    from oracle_lib import Oracle
    sb0 = ModelBlob.load('bed_settle')
    w = sb0.words.copy()
    for c in range(*sb0.meta['ranges']['bed']):
        w.view(np.float32)[sb0.h['OFF_COLL'] + c * L.C['STRIDE'] + L.C['FRICTION']] = 1.0
    sb = ModelBlob(w, sb0.meta)
    s = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'settle_forearm_on_mattress_edge.npy'))
    st = Stepper(sb, 2)
    lay = st.debug_layout()
    dbg = torch.zeros(2, lay[0], device='cuda')
    st.set_state(np.stack([s, s]))
    st.settle_debug(1, dbg)
    torch.cuda.synchronize()
    got = st.get_state()[0]
    D = dbg.cpu().numpy()[0]
    nc = int(D[0])
    ce = D[lay[1]:lay[1] + 1024].reshape(64, 16)[:nc]
    ref = s.copy()
    con = Oracle(sb).substep_debug(ref)
    assert [(int(x), int(y)) for x, y in ce.view(np.int32)[:, :2]] == [(int(c[0]), int(c[1])) for c in con]
    assert (3, 97) in [(int(c[0]), int(c[1])) for c in con]
    assert np.abs(ce[:, 13] - con[:, 11]).max() < 1e-5 and np.abs(ce[:, 10:13] - con[:, 8:11]).max() < 1e-3
    assert np.abs(got[:47] - ref[:47]).max() < 2e-5
