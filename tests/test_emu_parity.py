"""The PRODUCT kernel sources (assistive_gym_amd/csrc/agx_step.h, f32) compiled for the CPU wave
emulator (tests/emu/) against the f64 oracle: catches logic errors and non-uniform collectives
without a GPU.  The GPU run of the same comparison is tests/test_gpu_parity.py."""
import numpy as np
import pytest

from assistive_gym_amd.host.reset import make_states


@pytest.fixture(scope='module')
def emu(blob):
    from emu_lib import Emu
    return Emu(blob.set_param('NITER', 12))


@pytest.fixture(scope='module')
def oracle12(blob):
    from oracle_lib import Oracle
    return Oracle(blob.set_param('NITER', 12))


def test_settle_and_step_match(blob, emu, oracle12):
    st, _ = make_states(blob, 2, seed=3001)
    rng = np.random.RandomState(2)
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        oracle12.settle(so, 4); emu.settle(se, 4)
        assert np.abs(blob.view(so)['q'] - blob.view(se)['q']).max() < 1e-5
        # the 4 settle substeps contain the bowl's landing on the table (0.6 m/s, up to 28 candidate contacts on flat faces,
        # where f32 and f64 GJK may stop at different, equally close witness points): 3e-4 m there, 1e-6 elsewhere
        dp = np.abs(blob.view(so)['free'][0, :, :3] - blob.view(se)['free'][0, :, :3])
        assert dp.max() < 3e-4 and np.delete(dp, 1, axis=0).max() < 1e-5
        s = so
        for k in range(2):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            s1, s2 = s.copy(), s.copy()
            o_obs, o_rew, o_done, o_info = oracle12.step(s1, a)
            e_obs, e_rew, e_done, e_info, dbg = emu.step(s2, a, debug=True)
            assert o_info[6] == e_info[6] and o_info[7] == e_info[7]          # same contacts, same rows
            assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4
            assert np.abs(blob.view(s1)['q'] - blob.view(s2)['q']).max() < 1e-5
            s = s1


def test_observe_matches(blob, emu, oracle12):
    st, _ = make_states(blob, 1, seed=3005)
    assert np.abs(emu.observe(st[0].copy()) - oracle12.observe(st[0].copy())).max() < 1e-5


def test_tremor_step_matches(blob, emu, oracle12):
    """Dynamic head chain (tremor impairment): second articulated body, per-gender link records."""
    st, infos = make_states(blob, 2, seed=3101, impairment='tremor')
    rng = np.random.RandomState(4)
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        oracle12.settle(so, 3); emu.settle(se, 3)
        assert np.abs(blob.view(so)['q'] - blob.view(se)['q']).max() < 1e-5
        for k in range(2):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            s1, s2 = so.copy(), so.copy()
            o_obs, o_rew, o_done, o_info = oracle12.step(s1, a)
            e_obs, e_rew, e_done, e_info, _ = emu.step(s2, a)
            # the particle pile is chaotic at 12 sweeps: the last-substep contact set may differ by a few
            # near-tied candidates; robot, head and reward must still agree
            assert abs(o_info[6] - e_info[6]) <= 4
            assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4
            assert np.abs(blob.view(s1)['q'] - blob.view(s2)['q']).max() < 1e-5
            assert np.abs(blob.view(s1)['qt'] - blob.view(s2)['qt']).max() < 1e-6
            so = s1


def test_food_events_match(blob, emu, oracle12):
    """Finish kernel state machine (feeding.py:50-83): a particle teleported away from the spoon is
    a spill (-5, no longer alive), one placed at the mouth target is eaten (+20, task_success)."""
    st, _ = make_states(blob, 1, seed=3201)
    s0 = st[0].copy()
    oracle12.settle(s0, 3)
    v = blob.view(s0)
    food0 = blob.h['FOOD0']
    # displaced diagonally: every per-axis box gap to the spoon stays below the 0.1 m query distance but
    # the true distance exceeds it, so the spill decision rests on the narrowphase, not on the box test
    v['free'][0, food0 + 0, :3] += np.array([-0.095, -0.095, 0.095], dtype=np.float32)
    v['free'][0, food0 + 0, 7:13] = 0.0
    # released 4.5 cm above the mouth target with zero velocity: after the 0.1 s of free fall of one
    # env.step it is within the 3 cm mouth radius
    v['free'][0, food0 + 1, :3] = v['target'][0] + np.array([0.0, 0.0, 0.045], dtype=np.float32)
    v['free'][0, food0 + 1, 7:13] = 0.0
    a = np.zeros(blob.act_dim, dtype=np.float32)
    s1, s2 = s0.copy(), s0.copy()
    o_obs, o_rew, o_done, o_info = oracle12.step(s1, a)
    e_obs, e_rew, e_done, e_info, _ = emu.step(s2, a)
    v1, v2 = blob.view(s1), blob.view(s2)
    assert int(v1['food_alive'][0]) == int(v2['food_alive'][0]) and int(v1['food_active'][0]) == int(v2['food_active'][0])
    assert int(v1['food_alive'][0]) & 3 == 0                   # both particles left the alive set
    assert int(v1['task_success'][0]) == int(v2['task_success'][0]) == 1
    assert abs(o_rew - e_rew) < 1e-3 and np.abs(o_obs - e_obs).max() < 1e-4
    # the remaining particles are still on the spoon in both
    assert bin(int(v2['food_alive'][0])).count('1') == blob.nfood - 2


def test_edge_contact_is_not_lost(blob):
    """Regression: robot link resting on the table EDGE (tests/golden/edge_contact_state.npy, captured
    from a rollout).  The f32 GJK used to accept a degenerate last sub-simplex solve that moved away
    from the closest point and so dropped the contact; no-progress iterations now keep the best point."""
    import os
    from emu_lib import Emu
    from oracle_lib import Oracle
    s = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'edge_contact_state.npy'))
    b1 = blob.set_param('FRAME_SKIP', 1)
    con = Oracle(blob).substep_debug(s.copy())
    pairs_o = [(int(c[0]), int(c[1])) for c in con]
    emu = Emu(b1)
    dbg = emu.step(s.copy(), np.zeros(blob.act_dim, dtype=np.float32), debug=True)[4]
    nc = int(dbg[0])
    ce = dbg[emu.DBG_CON:emu.DBG_CON + 1024].reshape(64, 16)[:nc]
    pairs_e = [(int(x), int(y)) for x, y in ce.view(np.int32)[:, :2]]
    robot_o = [p for p in pairs_o if p[0] < 13]
    assert robot_o, 'fixture must contain a robot contact'
    assert pairs_e == pairs_o
    k = pairs_o.index(robot_o[0])
    assert abs(ce[k, 13] - con[k][11]) < 1e-5 and np.abs(ce[k, 10:13] - np.array(con[k][8:11])).max() < 1e-3


@pytest.mark.parametrize('impairment', ['limits', 'tremor'])
def test_coop_step_matches(blob, impairment):
    """Co-op flavour (FeedingJacoHumanEnv, feeding_envs.py:64-67): the human's head joints take actions
    (env.py:201-215, with the tremor branch), their limits are scaled per environment, and the
    observation carries the human's 23 values (feeding.py:102-108)."""
    from emu_lib import Emu
    from oracle_lib import Oracle
    co = blob.coop().set_param('NITER', 12)
    assert co.is_coop and co.act_dim == 11 and co.obs_dim == 48
    emu, orc = Emu(co), Oracle(co)
    st, infos = make_states(co, 1, seed=3301, impairment=impairment)
    v = co.view(st[0])
    assert int(v['frozen'][0]) == 0                                   # controllable joints are dynamic
    assert (float(v['limit_scale'][0]) < 1.0) == (impairment == 'limits')
    rng = np.random.RandomState(6)
    so, se = st[0].copy(), st[0].copy()
    orc.settle(so, 3); emu.settle(se, 3)
    moved = 0.0
    for k in range(3):
        a = rng.uniform(-1, 1, co.act_dim).astype(np.float32)
        a[7:] = np.sign(a[7:])                                        # drive the head joints hard, into their limits
        s1, s2 = so.copy(), so.copy()
        o_obs, o_rew, o_done, o_info = orc.step(s1, a)
        e_obs, e_rew, e_done, e_info, _ = emu.step(s2, a)
        assert o_obs.shape == (48,) and np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4
        v1, v2 = co.view(s1), co.view(s2)
        assert np.abs(v1['q'] - v2['q']).max() < 1e-5 and np.abs(v1['qt'] - v2['qt']).max() < 1e-6
        assert np.abs(v1['tremor_target'] - v2['tremor_target']).max() < 1e-6
        moved = max(moved, np.abs(v1['q'][0, co.nrobot:] - co.view(so)['q'][0, co.nrobot:]).max())
        # human joint angles appear raw in the human observation (feeding.py:103,108)
        assert np.abs(o_obs[25 + 10:25 + 14] - v1['q'][0, co.nrobot:]).max() < 1e-6
        lo = np.array([co.robot_f(d, 'LOWER', gender=int(v1['gender'][0])) for d in range(co.nrobot, co.ndof)]) * float(v1['limit_scale'][0])
        hi = np.array([co.robot_f(d, 'UPPER', gender=int(v1['gender'][0])) for d in range(co.nrobot, co.ndof)]) * float(v1['limit_scale'][0])
        assert (v1['q'][0, co.nrobot:] >= lo - 1e-6).all() and (v1['q'][0, co.nrobot:] <= hi + 1e-6).all()
        so = s1
    assert moved > 1e-3                                               # the head really is actuated


@pytest.mark.parametrize('param,value', [('MAX_CONTACTS', 12), ('MAX_ROWS', 60), ('MAX_ENTRIES', 500)])
def test_budget_truncation_matches(blob, param, value):
    """Edge case: the contact / row / coefficient budgets are exhausted.  Both implementations keep the
    same prefix of the contact list (collide: first MAX_CONTACTS selections; build_rows: largest prefix that
    fits MAX_ROWS and MAX_ENTRIES) and report the overflow."""
    from emu_lib import Emu
    from oracle_lib import Oracle
    b = blob.set_param('NITER', 8).set_param(param, value)
    emu, orc = Emu(b), Oracle(b)
    st, _ = make_states(b, 1, seed=3401)
    so, se = st[0].copy(), st[0].copy()
    orc.settle(so, 2); emu.settle(se, 2)
    a = np.random.RandomState(8).uniform(-1, 1, b.act_dim).astype(np.float32)
    o_obs, o_rew, o_done, o_info = orc.step(so, a)
    e_obs, e_rew, e_done, e_info, _ = emu.step(se, a)
    assert o_info[6] == e_info[6] and o_info[7] == e_info[7]                 # contacts and rows after truncation
    if param == 'MAX_CONTACTS':
        assert o_info[6] <= value
    if param == 'MAX_ROWS':
        assert o_info[7] <= value
    assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-3
    assert np.abs(b.view(so)['q'] - b.view(se)['q']).max() < 1e-5


def test_action_clipping_and_zero_action(blob, emu, oracle12):
    """take_step clips to the action space before scaling (env.py:187-188): +5 behaves like +1, and a zero
    action keeps the targets where the joints are."""
    st, _ = make_states(blob, 1, seed=3402)
    s0 = st[0].copy(); oracle12.settle(s0, 2)
    big, one = np.full(blob.act_dim, 5.0, np.float32), np.ones(blob.act_dim, np.float32)
    sa, sb = s0.copy(), s0.copy()
    ra = emu.step(sa, big); rb = emu.step(sb, one)
    assert np.array_equal(blob.view(sa)['qt'], blob.view(sb)['qt']) and np.array_equal(blob.view(sa)['q'], blob.view(sb)['q'])
    assert ra[1] < rb[1]                                                     # only the action penalty differs (feeding.py:27)
    sz = s0.copy(); emu.step(sz, np.zeros(blob.act_dim, np.float32))
    arm = [d for d in range(blob.nrobot) if blob.robot_i(d, 'ACT') >= 0]
    assert np.abs(blob.view(sz)['qt'][0, arm] - blob.view(s0)['q'][0, arm]).max() < 1e-6


def test_resting_bowl_face_manifold(blob, oracle):
    """A bowl at rest on the table is the tie-prone case of the face manifold (several coplanar lowest vertices per hull
    piece): the device code (f32) and the oracle (f64) must pick the same support points -- bowl pose / velocity after a
    step agree, and the bowl stays at rest."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from emu_lib import Emu
    e = Emu(blob)
    st, _ = ro.with_collision_check(blob.words).sample(41)
    oracle.settle(st, 25)
    rng = np.random.RandomState(3)
    for k in range(30):
        oracle.step(st, rng.uniform(-1, 1, blob.act_dim).astype(np.float32))
    so, se = st.copy(), st.copy()
    a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
    o_obs = oracle.step(so, a)[0]
    e_obs, _, _, e_info, _ = e.step(se, a)
    vo, ve = blob.view(so[None]), blob.view(se[None])
    assert np.abs(o_obs - e_obs).max() < 1e-5
    assert np.abs(vo['free'][0, 1, :7] - ve['free'][0, 1, :7]).max() < 1e-5
    assert np.abs(vo['free'][0, 1, 7:] - ve['free'][0, 1, 7:]).max() < 5e-4
    assert np.linalg.norm(ve['free'][0, 1, 10:13]) < 0.02 and np.linalg.norm(vo['free'][0, 1, 10:13]) < 0.02


def test_analytic_scenarios_on_the_device_code(blob, oracle):
    """The constructed cases of tests/test_oracle.py (free flight, head-on collision of two spheres, a sliding sphere
    that starts rolling) run through the kernel sources: same particle states as the oracle, which is itself checked
    against the closed forms there."""
    from emu_lib import Emu
    e = Emu(blob)
    st, _ = make_states(blob, 1, seed=2001)
    f0 = blob.h['FOOD0']; r = blob.free_f(f0, 'RADIUS')

    def scenario(setup, n):
        s = st[0].copy(); setup(blob.view(s))
        so, se = s.copy(), s.copy()
        oracle.settle(so, n); e.settle(se, n)
        fo, fe = blob.view(so)['free'][0, f0:f0 + 2], blob.view(se)['free'][0, f0:f0 + 2]
        assert np.abs(fo[:, :3] - fe[:, :3]).max() < 2e-6 and np.abs(fo[:, 7:10] - fe[:, 7:10]).max() < 2e-5, (fo, fe)
        assert np.abs(fo[:, 10:13] - fe[:, 10:13]).max() < 5e-3 * max(1.0, np.abs(fo[:, 10:13]).max())
        return fe

    def flight(v):
        v['free'][0, f0, :3], v['free'][0, f0, 7:13] = [1.5, 1.5, 2.5], [0.3, -0.2, 0.5, 0, 0, 0]
    scenario(flight, 4)

    def collide(v):
        c = np.array([1.5, 1.5, 2.5])
        v['free'][0, f0, :3], v['free'][0, f0 + 1, :3] = c - [0.02, 0, 0], c + [0.02, 0, 0]
        v['free'][0, f0, 7:13], v['free'][0, f0 + 1, 7:13] = [0.5, 0, 0, 0, 0, 0], [-0.5, 0, 0, 0, 0, 0]
    fe = scenario(collide, 3)
    assert abs(fe[0, 7]) < 1e-4 and abs(fe[1, 7]) < 1e-4 and abs(np.linalg.norm(fe[1, :3] - fe[0, :3]) - 2 * r) < 1e-5

    def slide(v):
        v['free'][0, f0, :3], v['free'][0, f0, 7:13] = [0.25, -1.0, 0.725 + r], [0.2, 0, 0, 0, 0, 0]
    fe = scenario(slide, 2)
    assert fe[0, 11] == pytest.approx(fe[0, 7] / r, rel=1e-2)             # rolling without slipping


def test_golden_trajectory_replay(blob):
    """The committed oracle trajectory (tests/golden/feeding_jaco_oracle_traj.npz: settled state, 20 actions, the oracle's
    observations / rewards / final state) replayed FREE-RUNNING through the kernel sources."""
    import os
    from emu_lib import Emu
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeding_jaco_oracle_traj.npz'))
    e = Emu(blob)
    s = g['state0'].copy()
    full = bool(os.environ.get('AGX_FULL_TESTS'))             # the emulator takes ~2 s per step: 4 of the 20 steps by default (the GPU suite replays all)
    for k in range(len(g['actions']) if full else 4):
        obs, rew, done, info, _ = e.step(s, g['actions'][k])
        assert np.abs(obs - g['obs'][k]).max() < 2e-4 and abs(rew - float(g['reward'][k])) < 2e-4, k
    if not full:
        return
    v, w = blob.view(s[None]), blob.view(g['state_end'][None].copy())
    assert np.abs(v['q'] - w['q']).max() < 1e-4 and np.abs(v['free'][0, :, :3] - w['free'][0, :, :3]).max() < 3e-3   # the particles jostle on the spoon: mm-level after 20 free-running steps
    assert v['food_alive'][0] == w['food_alive'][0] and v['iteration'][0] == w['iteration'][0]


@pytest.mark.parametrize('case', [0, 1, 2])
def test_pushed_bowl_and_particles(blob, oracle, case):
    """Cases of the kind a randomised emulator-vs-oracle campaign flagged before the first contact of a face pair was
    re-anchored at a vertex: the bowl (or a particle) is given a sudden velocity, slides / tips / lands again within the
    step.  With GJK's witness point on a flat face as first contact, f32 and f64 built different support polygons (bowl
    position off by up to 2.5 mm after one step, different contact counts); now all bodies agree."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from emu_lib import Emu
    rng = np.random.RandomState(77 + case)
    e = Emu(blob)
    st, _ = ro.with_collision_check(blob.words).sample(int(rng.randint(1, 1 << 30)))
    oracle.settle(st, 25)
    for k in range(int(rng.randint(5, 30))):
        oracle.step(st, rng.uniform(-1, 1, blob.act_dim).astype(np.float32))
    v = blob.view(st[None])
    v['free'][0, 1, 7:10] += rng.uniform(-0.3, 0.3, 3)                         # push the bowl
    v['free'][0, 2 + rng.randint(8), 7:10] += rng.uniform(-0.5, 0.5, 3)        # and a particle
    a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
    so, se = st.copy(), st.copy()
    oo = oracle.step(so, a)
    eo = e.step(se, a)
    vo, ve = blob.view(so[None]), blob.view(se[None])
    assert np.abs(oo[0] - eo[0]).max() < 1e-5 and abs(oo[1] - eo[1]) < 1e-5
    assert np.abs(vo['q'] - ve['q']).max() < 1e-5
    assert np.abs(vo['free'][0, :, :3] - ve['free'][0, :, :3]).max() < 5e-5
    assert oo[3][6] == eo[3][6] and oo[3][7] == eo[3][7]                        # same contacts and rows in the last substep


def test_nonfinite_state_is_flagged_and_masked(blob, emu):
    """SURVEY 5 robustness: an environment whose state has gone NaN reports done with a zeroed observation / reward and the
    AGX_INFO_NONFINITE marker, so that the auto-reset replaces it and a training batch is not poisoned (kernel sources on the emulator)"""
    st, _ = make_states(blob, 1, seed=3401)
    s = st[0].copy()
    v = blob.view(s)
    v['q'][0, 2] = np.nan
    obs, rew, done, info, _ = emu.step(s, np.zeros(blob.act_dim, dtype=np.float32))
    assert done and rew == 0.0 and np.all(obs == 0.0) and info[6] == 1.0e6
    s = st[0].copy()                                             # a healthy state is left alone
    obs, rew, done, info, _ = emu.step(s, np.zeros(blob.act_dim, dtype=np.float32))
    assert not done and np.isfinite(obs).all() and info[6] < 1.0e6


def test_noop_retest_rule_against_the_plain_solve(blob):
    """The no-op re-test rule (AGX_P_NOOP_RETEST = 5) is an approximation the oracle shares; here the kernel sources WITH the rule (the full 50
    sweeps) against the ORACLE WITHOUT it (NOOP_RETEST = 0, the plain solve): settled FeedingJaco states, three steps each, reward / forces /
    observation to 1e-3 relative.  The same comparison on 64 environments x 20 steps of configs 2 and 3 (wiping) runs on the GPU
    (tests/test_gpu_parity.py); the sensitivity study behind the rule is tests/diag/noop_retest_sensitivity.py."""
    from emu_lib import Emu
    from oracle_lib import Oracle
    assert blob.param('NOOP_RETEST') == 5 and blob.param('NITER') == 50
    e, with_rule, plain = Emu(blob), Oracle(blob), Oracle(blob.set_param('NOOP_RETEST', 0.0))
    st, _ = make_states(blob, 3, seed=3601)
    rng = np.random.RandomState(6)
    differs = 0
    for i in range(3):
        s = st[i].copy()
        with_rule.settle(s, 25)
        for k in range(3):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            s_e, s_p, s_r = s.copy(), s.copy(), s.copy()
            obs, rew, done, info, _ = e.step(s_e, a)
            o_obs, o_rew, o_done, o_info = plain.step(s_p, a)
            r_obs, r_rew, _, _ = with_rule.step(s_r, a)
            differs += int(not np.array_equal(s_p, s_r))
            assert np.abs(obs - o_obs).max() < 1e-3 and abs(rew - o_rew) <= 1e-3 * max(1.0, abs(o_rew)) and abs(info[0] - o_info[0]) <= 1e-3 * max(1.0, abs(o_info[0])), (i, k)
            vq, vp = blob.view(s_e[None])['q'][0], blob.view(s_p[None])['q'][0]
            assert np.abs(vq - vp).max() < 1e-4
            s = s_p
    assert differs > 0                                     # (the rule does change the arithmetic: a real comparison)


@pytest.mark.parametrize('path', ['velocity_space', 'row_space'])
def test_warm_start_switch_matches_the_oracle(blob, path):
    """AGX_P_WARMSTART (a [BULLET-UNVERIFIED] convention: SOLVER_USE_WARMSTARTING, factor 0.85; default off) on the kernel sources: the solve
    kernel leaves (contact key, impulse) in the scratch record, the next build kernel seeds the contact normals, the sweeps start from them --
    the velocity-space sweep of FeedingJaco (food pile: dozens of resting contacts) and the row-space sweep of BedBathingSawyer (pad pressed
    on the arm).  Against the oracle's switch over four consecutive steps (the memory persists from step to step on both sides), and the
    switch must change the result (a real comparison)."""
    import os, sys
    from emu_lib import Emu
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    if path == 'velocity_space':
        b0 = blob
        st, _ = make_states(b0, 1, seed=3001)
        s = st[0].copy(); Oracle(b0).settle(s, 25)
        scale = 1.0
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b0 = ModelBlob.load('bed_bathing_sawyer')
        s = wiping_pool(b0, 2, 6006)[1].copy(); b0.view(s[None])['iteration'][0] = 0
        scale = 0.15
    b = b0.set_param('WARMSTART', 0.85)
    o, e, cold = Oracle(b), Emu(b), Oracle(b0)
    o.forget_warm(); e.forget_warm()
    so, se, sc = s.copy(), s.copy(), s.copy()
    rng = np.random.RandomState(5)
    differs = 0.0
    for k in range(4):
        a = (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32)
        o_obs, o_rew, _, o_info = o.step(so, a)
        obs, rew, _, info, _ = e.step(se, a)
        c_obs, c_rew, _, c_info = cold.step(sc, a)
        assert np.abs(obs - o_obs).max() < 2e-5 and abs(rew - o_rew) < 2e-5 * max(1.0, abs(o_rew)) and abs(info[0] - o_info[0]) <= 1e-3 * max(1.0, abs(o_info[0])), (path, k)
        assert np.abs(b.view(se[None])['q'][0] - b.view(so[None])['q'][0]).max() < 5e-6
        differs = max(differs, float(np.abs(so - sc)[:b.h['S_ENV']].max()))
        se[:] = so; sc[:] = so
    # (the food pile's 50 sweeps are not converged: the start matters; the pad's small system is converged to the last bit in 50 sweeps either way)
    assert differs > 1e-6 or path == 'row_space'
    o.forget_warm()


def test_relative_travel_bound_leaves_the_contacts_unchanged(blob):
    """rel_travel() (csrc/agx_collide.h) only drops pairs that cannot produce a solver row: the kernel sources with and without it
    (-DAGX_NO_REL_TRAVEL) produce bit-identical state records over random-policy steps of settled FeedingJaco states."""
    from emu_lib import Emu
    from oracle_lib import Oracle
    st, _ = make_states(blob, 2, seed=5151)
    new, old = Emu(blob), Emu(blob, kind='feeding_abs_travel')
    o = Oracle(blob)
    rng = np.random.RandomState(9)
    for i in range(len(st)):
        o.settle(st[i], 25)
        s1, s2 = st[i].copy(), st[i].copy()
        for k in range(3):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            r1 = new.step(s1, a); r2 = old.step(s2, a)
            assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32)), (i, k)
            assert np.array_equal(r1[0], r2[0]) and r1[1] == r2[1]


@pytest.mark.parametrize('path', ['velocity_space', 'row_space'])
def test_second_friction_direction_matches_the_oracle(blob, path):
    """AGX_P_FRICTION_DIRS = 2 (a [BULLET-UNVERIFIED] convention: SOLVER_USE_2_FRICTION_DIRECTIONS; default off) on the kernel sources: a
    second block of friction rows along n x t behind the first, each bounded by mu x the normal impulse.  The velocity-space sweep of
    FeedingJaco re-reads the block's row sets per sweep, the row-space sweep of BedBathingSawyer just has more rows (an environment whose
    rows no longer fit its 56-row work area falls back to the velocity-space sweep, as always).  Against the oracle's switch over four
    steps; the switch must change the result."""
    import os, sys
    from emu_lib import Emu
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    if path == 'velocity_space':
        b0 = blob
        st, _ = make_states(b0, 1, seed=3001)
        s = st[0].copy(); Oracle(b0).settle(s, 25)
        scale = 1.0
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b0 = ModelBlob.load('bed_bathing_sawyer')
        s = wiping_pool(b0, 2, 6006)[1].copy(); b0.view(s[None])['iteration'][0] = 0
        scale = 0.15
    b = b0.set_param('FRICTION_DIRS', 2.0)
    o, e, one = Oracle(b), Emu(b), Oracle(b0)
    so, se, sc = s.copy(), s.copy(), s.copy()
    rng = np.random.RandomState(5)
    differs, rows = 0.0, []
    for k in range(4):
        a = (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32)
        o_obs, o_rew, _, o_info = o.step(so, a)
        obs, rew, _, info, _ = e.step(se, a)
        c_obs, c_rew, _, c_info = one.step(sc, a)
        assert info[7] == o_info[7] and info[6] == o_info[6], (path, k, info, o_info)            # rows and contacts of the last substep
        rows.append((int(o_info[7]), int(o_info[6]), int(c_info[7]), int(c_info[6])))
        assert np.abs(obs - o_obs).max() < 2e-5 and abs(rew - o_rew) < 2e-5 * max(1.0, abs(o_rew)) and abs(info[0] - o_info[0]) <= 1e-3 * max(1.0, abs(o_info[0])), (path, k)
        assert np.abs(b.view(se[None])['q'][0] - b.view(so[None])['q'][0]).max() < 5e-6
        differs = max(differs, float(np.abs(so - sc)[:b.h['S_ENV']].max()))
        se[:] = so; sc[:] = so
    # three rows per contact instead of two next to the same non-contact rows (FeedingJaco's 160-row budget then holds 47 contacts, not 53)
    assert all(r2 - 3 * n2 == r1 - 2 * n1 for r2, n2, r1, n1 in rows[:1]) and any(n2 > 0 for _, n2, _, _ in rows), rows
    assert differs > 1e-7, differs


@pytest.mark.parametrize('scene', ['feeding', 'wiping', 'scratching'])
def test_persistent_manifold_matches_the_oracle(blob, scene):
    """AGX_P_MANIFOLD (a [BULLET-UNVERIFIED] convention: btPersistentManifold, default off) on the kernel sources: cached contact points of the
    hull pairs live across substeps and steps in the scratch record -- refreshed, merged with the substep's GJK contacts (nearest replaces,
    append, or the area rule when four are cached), and the contact list is rebuilt from them.  Against the oracle's switch over consecutive
    steps (the memory persists on both sides); contact and row counts must agree exactly, and the switch must change the result."""
    import os, sys
    from emu_lib import Emu
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    steps = 4
    if scene == 'feeding':
        b0 = blob
        st, _ = make_states(b0, 1, seed=3001)
        s = st[0].copy(); Oracle(b0).settle(s, 25)
        scale = 1.0
    elif scene == 'wiping':
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import wiping_pool
        b0 = ModelBlob.load('bed_bathing_sawyer')
        s = wiping_pool(b0, 4, 6006)[3].copy(); b0.view(s[None])['iteration'][0] = 0
        scale = 0.15
    else:
        d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scratch_sawyer_parallel_edge_case.npz'))
        b0 = ModelBlob.load('scratch_itch_sawyer'); s = d['start'].copy(); scale = 0.1
    b = b0.set_param('MANIFOLD', 1.0)
    o, e, plain = Oracle(b), Emu(b), Oracle(b0)
    o.forget_warm(); e.forget_warm(); o.manifold_stats()
    so, se, sp = s.copy(), s.copy(), s.copy()
    rng = np.random.RandomState(5)
    differs, more = 0.0, 0
    f = b.obs_dim_robot - 1
    for k in range(steps):
        a = (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32)
        o_obs, o_rew, _, o_info = o.step(so, a)
        obs, rew, _, info, _ = e.step(se, a)
        p_info = plain.step(sp, a)[3]
        assert info[6] == o_info[6] and info[7] == o_info[7], (scene, k, info, o_info)
        more += int(o_info[6] > p_info[6])
        tol = 2e-5 if scene != 'feeding' else 1e-4
        assert np.abs(np.delete(obs - o_obs, f)).max() < tol and abs(rew - o_rew) < tol * max(1.0, abs(o_rew)) + 0.06 * 1e-2, (scene, k, np.abs(obs - o_obs).max())
        assert abs(obs[f] - o_obs[f]) <= max(1e-3 * max(1.0, abs(o_obs[f])), 1e-2) and abs(info[0] - o_info[0]) <= max(1e-3 * max(1.0, abs(o_info[0])), 1e-2), (scene, k, obs[f], o_obs[f])
        differs = max(differs, float(np.abs(so - sp)[:b.h['S_ENV']].max()))
        se[:] = so; sp[:] = so
    # (the wiping and scratching steps end with as many contacts, at other points: tool x person keeps every contact inside the break distance since round 5, group flag bit 6)
    assert differs > 1e-7 and (more > 0 or scene != 'feeding'), (differs, more)
    stats = o.manifold_stats()
    print('%s: manifold points replaced %d, appended %d, replaced by the area rule %d, dropped at the refresh %d' % ((scene,) + tuple(int(x) for x in stats)))
    assert stats[0] > 0 and stats[1] > 0 and (scene != 'wiping' or stats[3] > 0), stats          # (the pad slides: points drift out and are dropped)
    o.forget_warm()


def test_persistent_manifold_area_rule():
    """The fifth point of a pair (sortCachedPoints): colliders of the scenes here are smaller than the 2 cm within which a new point REPLACES its
    nearest cached neighbour, so the rule is reached through the test hooks -- four cached points of the scratcher-on-forearm pair placed 3 cm
    around the real contact, at its distance and normal (they pass the refresh: no drift), then one step.  Oracle and kernel sources must
    end with the same five-minus-one points, and the rule must have fired."""
    import os
    from emu_lib import Emu
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scratch_sawyer_parallel_edge_case.npz'))
    b = ModelBlob.load('scratch_itch_sawyer').set_param('MANIFOLD', 1.0).set_param('FRAME_SKIP', 1.0)
    o, e = Oracle(b), Emu(b)
    zero = np.zeros(b.act_dim, dtype=np.float32)
    o.forget_warm(); e.forget_warm(); o.manifold_stats()
    s0 = d['start'].copy()
    so = s0.copy(); o.step(so, zero)
    real = o.manifold_get()
    assert len(real) >= 1
    p = real[0]
    # four more points of the pair: 3 cm from the real one along the local x / y axes of A, each with the point of B that lies at the real
    # contact's distance along its normal (no drift at the refresh) -- B's frame (a link of the human's arm) from the oracle's kinematics, A's
    # (the tool, a free body) from the record, both AFTER the step: the refresh of the next substep sees those poses
    from assistive_gym_amd.model import xform as X
    from assistive_gym_amd.model import compiler as L
    ca, cb = int(p[0]), int(p[1])
    ba, bb = b.collider(ca)['body'], b.collider(cb)['body']
    assert L.BODY_FREE0 <= ba < L.BODY_HUMAN0 and 0 <= bb < L.BODY_ROBOT_BASE
    vo = b.view(so.reshape(1, -1))
    pa_, qa_ = vo['free'][0, ba - L.BODY_FREE0, :3].astype(np.float64), vo['free'][0, ba - L.BODY_FREE0, 3:7].astype(np.float64)
    Ra = X.quat_to_mat(qa_)
    lpos, lrot = o.fk(so)
    pb_, Rb = lpos[bb], lrot[bb]
    n = p[8:11]
    dist0 = float((Ra @ p[2:5] + pa_ - (Rb @ p[5:8] + pb_)) @ n)
    rows = []
    for off in ([0.03, 0, 0], [-0.03, 0, 0], [0, 0.03, 0], [0, -0.03, 0]):
        q = p.copy(); q[2:5] += off
        wa = Ra @ q[2:5] + pa_
        q[5:8] = Rb.T @ (wa - n * dist0 - pb_)
        rows.append(q)
    rows = np.array(rows)
    o.manifold_set(rows); e.manifold_set(rows); o.manifold_stats()
    s1, s2 = so.copy(), so.copy()
    o_out = o.step(s1, zero); e_out = e.step(s2, zero)
    stats = o.manifold_stats()
    assert stats[2] == 1, stats                                     # the area rule fired once
    mo, me = o.manifold_get(), e.manifold_get()
    # (other pairs of the tool with the arm lie inside the break distance and have cached points of their own since round 5: group flag bit 6)
    assert len(mo) == len(me) and np.array_equal(mo[:, :2], me[:, :2]) and int(((mo[:, 0] == ca) & (mo[:, 1] == cb)).sum()) == 4
    assert np.abs(mo[:, 2:] - me[:, 2:]).max() < 2e-5, np.abs(mo - me).max()
    assert e_out[3][6] == o_out[3][6] and np.abs(e_out[0] - o_out[0]).max() < 1e-3
    o.forget_warm()


def test_split_impulse_threshold_matches_the_oracle():
    """AGX_P_SPLIT_PEN (a [BULLET-UNVERIFIED] convention, default off; Bullet: 4 cm): a contact deeper than the threshold gets no positional
    term in its row.  The pressed scratcher of the committed fixture (0.3 mm inside the forearm) with a threshold of 0.2 mm: the kernel
    sources against the oracle's switch, and the contact force must drop against the default (the push-out term is gone)."""
    import os
    from emu_lib import Emu
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scratch_sawyer_parallel_edge_case.npz'))
    b0 = ModelBlob.load('scratch_itch_sawyer').set_param('FRAME_SKIP', 1.0)
    b = b0.set_param('SPLIT_PEN', 0.0002)
    f = b.obs_dim_robot - 1
    zero = np.zeros(b.act_dim, dtype=np.float32)
    s1, s2, s3 = d['start'].copy(), d['start'].copy(), d['start'].copy()
    o_obs = Oracle(b).step(s1, zero)[0]
    e_obs = Emu(b).step(s2, zero)[0]
    p_obs = Oracle(b0).step(s3, zero)[0]
    assert p_obs[f] > 1.0 and o_obs[f] < 0.8 * p_obs[f], (p_obs[f], o_obs[f])          # 2.25 N with the positional term, 1.60 N without
    assert np.abs(np.delete(e_obs - o_obs, f)).max() < 2e-5 and abs(e_obs[f] - o_obs[f]) <= 1e-3 * max(1.0, abs(o_obs[f]))
    assert np.abs(b.view(s1[None])['q'][0] - b.view(s2[None])['q'][0]).max() < 5e-6


def test_all_solver_switches_together_match_the_oracle(blob):
    """warm start + second friction direction + persistent manifold + split-impulse threshold at once (the combination a comparison with a
    PyBullet dump would start from): the kernel sources against the oracle over consecutive FeedingJaco steps, the memories of both sides
    (impulses, cached points) built up step by step."""
    from emu_lib import Emu
    from oracle_lib import Oracle
    st, _ = make_states(blob, 1, seed=3001)
    s = st[0].copy(); Oracle(blob).settle(s, 25)
    b = blob.set_param('WARMSTART', 0.85).set_param('FRICTION_DIRS', 2.0).set_param('MANIFOLD', 1.0).set_param('SPLIT_PEN', 0.04)
    o, e = Oracle(b), Emu(b)
    o.forget_warm(); e.forget_warm()
    so, se = s.copy(), s.copy()
    rng = np.random.RandomState(11)
    for k in range(3):
        a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
        o_obs, o_rew, _, o_info = o.step(so, a)
        obs, rew, _, info, _ = e.step(se, a)
        assert info[6] == o_info[6] and info[7] == o_info[7], (k, info, o_info)
        assert np.abs(obs - o_obs).max() < 1e-4 and abs(rew - o_rew) < 1e-4 * max(1.0, abs(o_rew)), (k, np.abs(obs - o_obs).max())
        se[:] = so
    o.forget_warm()


def test_row_local_sweep_against_the_register_sweep(blob):
    """The row-local sweep (csrc/agx_pgs_lv.h, -DAGX_PGS_LV=2: velocity deltas in LDS, lane = entry of the visited row -- on the device its
    visit loop is the assembly twin of the C++ run here) against the register sweep (csrc/agx_pgs.h, built with -DAGX_PGS_LV=0) and against itself with a 300-pair LDS window
    (most rows stream their pairs from the scratch record): same rows, same order, same clamps -- the dot products are associated
    differently, so the three agree to rounding, not bit for bit, and the window size must not change a bit."""
    from emu_lib import Emu
    from oracle_lib import Oracle
    lv, reg, cap, oracle = Emu(blob, 'feeding_lv2'), Emu(blob, 'feeding_reg'), Emu(blob, 'feeding_lv_cap'), Oracle(blob)
    st, _ = make_states(blob, 2, seed=3701)
    rng = np.random.RandomState(8)
    differs = 0
    for i in range(2):
        s = st[i].copy()
        oracle.settle(s, 20)
        for k in range(2):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            s_l, s_r, s_c, s_o = s.copy(), s.copy(), s.copy(), s.copy()
            l_obs, l_rew, _, l_info, _ = lv.step(s_l, a)
            r_obs, r_rew, _, r_info, _ = reg.step(s_r, a)
            c_obs, c_rew, _, c_info, _ = cap.step(s_c, a)
            o_obs, o_rew, _, o_info = oracle.step(s_o, a)
            assert np.array_equal(s_l, s_c) and np.array_equal(l_obs, c_obs) and l_rew == c_rew          # the window is a cache, not arithmetic
            differs += int(not np.array_equal(s_l, s_r))
            assert l_info[6] == r_info[6] == o_info[6] and l_info[7] == r_info[7] == o_info[7]         # same contacts, same rows
            vl, vr, vo = blob.view(s_l[None]), blob.view(s_r[None]), blob.view(s_o[None])
            for key, tol in (('q', 2e-6), ('qd', 2e-5)):
                assert np.abs(vl[key] - vr[key]).max() < tol and np.abs(vl[key] - vo[key]).max() < 5 * tol, (i, k, key)
            assert np.abs(vl['free'][0, :, :3] - vr['free'][0, :, :3]).max() < 2e-6
            assert np.abs(l_obs - o_obs).max() < 1e-4 and abs(l_rew - o_rew) < 1e-4 and abs(l_info[0] - o_info[0]) <= 1e-3 * max(1.0, abs(o_info[0]))
            s = s_o
    assert differs > 0                                     # (two different sweeps: a real comparison)


def test_wide_row_local_sweep_is_bit_identical(blob):
    """csrc/agx_pgs_lvw.h (the default solve path of the feeding variant since round 6: up to four rows with disjoint velocity slots per visit, one per
    16-lane group, list-scheduled per substep) against csrc/agx_pgs_lvs.h (one row per visit, the default of round 5): rows that share no slot
    commute exactly and rows that do keep their order, so states, observations, rewards and info agree BIT FOR BIT -- over settling (the bowl
    lands, food falls onto the spoon) and steps; also with a 300-pair LDS window (steps with a row beyond it read every pair from the scratch
    record), with the blob switch AGX_P_SOLVE_WIDE = 0 (the narrow sweep inside the wide build), and in a build whose scheduler gives up at 8
    steps per part (the fallback to the narrow sweep, substep by substep)."""
    from emu_lib import Emu
    wide, narrow, cap, few = Emu(blob), Emu(blob, 'feeding_lvs'), Emu(blob, 'feeding_lvw_cap'), Emu(blob, 'feeding_lvw_8steps')
    off = Emu(blob.set_param('SOLVE_WIDE', 0))
    st, _ = make_states(blob, 3, seed=3703)
    rng = np.random.RandomState(10)
    for i in range(3):
        s = st[i].copy()
        if i < 2:
            sw, sn = s.copy(), s.copy()
            wide.settle(sw, 6); narrow.settle(sn, 6)
            assert np.array_equal(sw.view(np.uint32), sn.view(np.uint32)), i
            s = sw
        for k in range(3):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            outs = []
            for e in (wide, narrow, cap, few, off):
                se = s.copy()
                obs, rew, done, info, _ = e.step(se, a)
                outs.append((se, obs, rew, info))
            for name, (se, obs, rew, info) in zip(('narrow', 'window of 300 pairs', 'scheduler limited to 8 steps', 'SOLVE_WIDE = 0'), outs[1:]):
                assert np.array_equal(se.view(np.uint32), outs[0][0].view(np.uint32)) and np.array_equal(obs, outs[0][1]) and rew == outs[0][2] and np.array_equal(info, outs[0][3]), (i, k, name)
            s = outs[0][0]


def test_row_local_sweep_with_scalar_headers(blob):
    """csrc/agx_pgs_lvs.h (the fallback of the wide sweep; the default of round 5: row headers through scalar loads from the scratch record, impulses in a vector register, velocity
    slots by arithmetic on the header) does the arithmetic of csrc/agx_pgs_lv.h visit by visit: the two agree BIT FOR BIT, and so does a build whose LDS
    window holds 300 pairs only (most rows read their pairs from the scratch record: the window is a cache, not arithmetic); the register sweep
    (-DAGX_PGS_LV=0) associates the dot products differently and agrees to rounding."""
    from emu_lib import Emu
    lv, lvs, cap, reg = Emu(blob, 'feeding_lv2'), Emu(blob, 'feeding_lvs'), Emu(blob, 'feeding_lvs_cap'), Emu(blob, 'feeding_reg')
    st, _ = make_states(blob, 2, seed=3702)
    rng = np.random.RandomState(9)
    differs = 0
    for i in range(2):
        s = st[i].copy()
        for k in range(3):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            outs = []
            for e in (lv, lvs, cap, reg):
                se = s.copy()
                obs, rew, done, info, _ = e.step(se, a)
                outs.append((se, obs, rew, info))
            (s_l, o_l, r_l, i_l), (s_s, o_s, r_s, i_s), (s_c, o_c, r_c, i_c), (s_r, o_r, r_r, i_r) = outs
            assert np.array_equal(s_l, s_s) and np.array_equal(o_l, o_s) and r_l == r_s and np.array_equal(i_l, i_s), (i, k)
            assert np.array_equal(s_c, s_s) and np.array_equal(o_c, o_s) and r_c == r_s, (i, k)
            differs += int(not np.array_equal(s_s, s_r))
            s = s_s
    assert differs > 0                                     # (a row-local sweep ran, not the register sweep)


def test_support_scan_rounds_are_bit_identical(blob):
    """csrc/agx_gjk.h gjk_support: vertex 0 inside the first round, eight vertices per round, one-vertex cores not scanned (AGX_GJK_SCAN_WIDE 2,
    AGX_GJK_SCAN_ONE 1: the default since the end of round 6) against the scan of rounds 3-5 (vertex 0, then four per round): the same comparisons in the
    same order, the winner re-loaded by index -- states, observations, rewards and info agree BIT FOR BIT over settling and steps (on the GPU:
    tests/test_gpu_solve_variants.py, and 13 task / robot combinations in profiles/r06/r06w_*, r06za_*)."""
    from emu_lib import Emu
    new, old = Emu(blob), Emu(blob, 'feeding_scan4')
    st, _ = make_states(blob, 3, seed=3811)
    rng = np.random.RandomState(12)
    for i in range(3):
        s = st[i].copy()
        sn, so = s.copy(), s.copy()
        new.settle(sn, 5); old.settle(so, 5)
        assert np.array_equal(sn.view(np.uint32), so.view(np.uint32)), i
        s = sn
        for k in range(3):
            a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
            s1, s0 = s.copy(), s.copy()
            o1, r1, d1, i1, _ = new.step(s1, a); o0, r0, d0, i0, _ = old.step(s0, a)
            assert np.array_equal(s1.view(np.uint32), s0.view(np.uint32)) and np.array_equal(o1, o0) and r1 == r0 and np.array_equal(i1, i0), (i, k)
            s = s1
