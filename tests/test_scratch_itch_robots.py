"""ScratchItch<Robot>-v1 for Jaco (the reference's default environment: env_viewer.py:36, learn.py:189), Panda and Sawyer (SURVEY 8 row
f3: more robots on the same kernels) without a GPU: blobs against the reference's robot tables, the host reset (mounted IK / base
search) with init_robot_pose's collision rejection driven by the device code's collision pass (here on the wave emulator), and the
scratch_itch kernel variant on the emulator against the oracle incl. a scratching contact.  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X
from test_scratch_itch import target_world, tip_pose

from conftest import full

ROBOTS = ['jaco', pytest.param('panda', marks=full), 'sawyer', pytest.param('baxter', marks=full)]


@pytest.fixture(scope='module', params=ROBOTS)
def rb(request):
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    from oracle_lib import Oracle
    b = ModelBlob.load('scratch_itch_' + request.param)
    return request.param, b, Oracle(b), Emu(b)


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset_scratch import make_states
    return make_states(blob, n, seed=seed, **kw)


def scratching_state(blob, oracle, seed, depth, checker=None):
    """a post-reset state whose ARM is moved (IK at fixed end-effector orientation; the base of a mounted robot cannot be shifted) so that
    the scratcher's tip (sphere r = 1 cm) presses `depth` into the skin at the target"""
    from assistive_gym_amd.host.reset_scratch import ScratchItchReset
    st, infos = _states(blob, 1, seed, checker=checker)
    s = st[0].copy()
    tgt, lp, lR = target_world(blob, oracle, s)
    axis = lR @ np.array([0, 0, -1.0])
    radial = (tgt - lp) - np.dot(tgt - lp, axis) * axis
    radial /= np.linalg.norm(radial)
    want = tgt + radial * (0.01 - depth)
    rs = ScratchItchReset(blob)
    v = blob.view(s.reshape(1, -1))
    bp, bR = v['base'][0, :3].astype(np.float64)[None], X.quat_to_mat(v['base'][0, 3:].astype(np.float64))[None]
    q = np.array([v['q'][0, d] for d in rs.arm.chain], dtype=np.float64)[None]
    pe0, Re0, _, _ = rs.arm.fk(bp, bR, q)
    tip0, _ = tip_pose(blob, s)
    q1 = rs.arm.ik(bp, bR, q, pe0 + (want - tip0)[None], Re0, iters=300, damping=0.01, maxstep=0.2)
    pe1, Re1, _, _ = rs.arm.fk(bp, bR, q1)
    assert np.linalg.norm(pe1[0] - pe0[0] - (want - tip0)) < 1e-4 and np.abs(Re1 - Re0).max() < 1e-3, 'the target is within reach'
    for k, d in enumerate(rs.arm.chain):
        v['q'][0, d] = v['qt'][0, d] = q1[0, k]
    v['free'][0, 0, :3] += (pe1[0] - pe0[0]).astype(np.float32)
    v['free'][0, 0, 7:] = 0
    return s


def emu_checker(emu):
    return lambda states: np.array([emu.check_collisions(s) for s in states], dtype=np.uint8)


def flags_from_oracle(blob, o, s):
    """the rule of csrc/agx_env.h collision_flags applied to the oracle's contact list"""
    f = 0
    for r in o.collide(s):
        ta, tb = blob.collider(int(r[0]))['tag'], blob.collider(int(r[1]))['tag']
        ra, rb_ = ta in (1, 2), tb in (1, 2)
        oa, ob = ta in (3, 6, 8, 9), tb in (3, 6, 8, 9)
        if ((ra and ob) or (rb_ and oa)) and r[11] <= 0:
            f |= 1
        if ra and rb_ and r[11] < -0.01:
            f |= 2
    return f


def test_model_tables(rb):
    from assistive_gym_amd.model import compiler as L
    name, b, o, e = rb
    T = L.SCRATCH_ROBOTS[name]
    assert b.task_kind == L.TASK_SCRATCH_ITCH and (b.act_dim, b.obs_dim, b.nhdof) == (7, 30, 10)               # scratch_itch.py:8: 23 + 7
    arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
    assert [b.robot_i(d, 'PB_INDEX') for d in arm_dofs] == T['arm']
    grip_dofs = [d for d in range(b.nrobot) if b.robot_i(d, 'PB_INDEX') in T['grip']]
    assert np.allclose([b.robot_f(d, 'QT0') for d in grip_dofs], T['gripper_target'])                          # gripper_pos['scratch_itch']
    assert np.isclose(b.robot_f(arm_dofs[0], 'KP'), 0.05) and np.isclose(b.robot_f(arm_dofs[0], 'MAXF'), 1.0)  # robot.py:36-37
    assert b.meta['mount'] == ('toc' if name in ('sawyer', 'baxter') else 'wheelchair')
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (17, 64)


def test_reset_places_the_tool_at_the_target_pose(rb):
    name, b, o, e = rb
    st, infos = _states(b, 4, 2001)
    want_q = X.quat_from_rpy(b.meta['ee_rpy'])
    for i in range(4):
        if infos[i]['toc_goals'] == 0:
            continue                                  # no IK solution within the threshold: the closest one is kept (robot.py:117-121)
        p, q = o.ee_pose(st[i])
        assert np.linalg.norm(p - infos[i]['target_ee_pos']) < 0.031                                           # robot.py:97 / env.py:297 thresholds
        assert min(np.linalg.norm(q - want_q), np.linalg.norm(q + want_q)) < 0.031
        v = b.view(st[i:i + 1])
        if b.meta['mount'] == 'wheelchair':           # scratch_itch.py:97-99: wheelchair position + toc_base_pos_offset, rpy (0, 0, -pi/2)
            assert np.allclose(v['base'][0, :3], np.array([0, 0, 0.06]) + b.meta['toc_base'], atol=1e-6)
            assert np.allclose(v['base'][0, 3:], X.quat_from_rpy([0, 0, -np.pi / 2.0]), atol=1e-6)
    assert sum(1 for i in infos if i['toc_goals'] > 0) >= 3


def test_collision_pass_matches_the_oracle_and_rejection_clears_the_pool(rb):
    name, b, o, e = rb
    n = 16
    raw, _ = _states(b, n, 3001)
    got = emu_checker(e)(raw)
    want = np.array([flags_from_oracle(b, o, s) for s in raw])
    assert np.array_equal(got, want)
    st, infos = _states(b, n, 3001, checker=emu_checker(e))
    after = np.array([flags_from_oracle(b, o, s) for s in st])
    assert np.array_equal(after, [i['collision_flags'] for i in infos])
    assert (after != 0).sum() <= max(1, (want != 0).sum() // 2), (want, after)       # re-draws clear (most of) the colliding placements
    keep = want == 0
    assert np.array_equal(st[keep], raw[keep])                                        # clean placements are not touched
    v0, v1 = b.view(raw), b.view(st)
    assert np.array_equal(v0['human'], v1['human']) and np.array_equal(v0['task'], v1['task'])     # only the robot is placed again


def test_emulator_matches_oracle_in_free_space_and_scratching(rb):
    name, b, o, e = rb
    st, infos = _states(b, 2, 4001, checker=emu_checker(e))
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        for k in range(3):
            a = np.random.RandomState(10 * i + k).uniform(-1, 1, 7).astype(np.float32)
            oo, orr, od, oi = o.step(so, a)
            eo, er, ed, ei, _ = e.step(se, a)
            assert oi[6] == ei[6] and abs(oi[7] - ei[7]) <= 0
            assert np.abs(oo - eo).max() < 2e-5 and abs(orr - er) < 2e-5
    s = scratching_state(b, o, seed=4101, depth=0.003, checker=emu_checker(e))
    assert not flags_from_oracle(b, o, s) & 2
    so, se = s.copy(), s.copy()
    hits = 0
    for k in range(3):
        a = (np.random.RandomState(k).uniform(-1, 1, 7) * 0.1).astype(np.float32)
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6]
        assert np.abs(oo[:-1] - eo[:-1]).max() < 1e-4 and abs(oo[-1] - eo[-1]) <= 1e-3 * max(1.0, abs(oo[-1]))
        for c in (0, 2, 3):
            assert abs(oi[c] - ei[c]) <= 1e-3 * max(1.0, abs(oi[c]))
        hits += int(oi[3] > 0)
    assert hits >= 1, 'the scratcher presses on the arm'


# ---- the device-side reset generator for the wheelchair-mounted arms (csrc/agx_reset.h on the wave emulator vs oracle/reset_oracle.py) --------
@pytest.mark.parametrize('robot', ['jaco', pytest.param('panda', marks=full)])
def test_device_reset_generator_matches_its_restatement(robot):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.model import compiler as L
    from emu_lib import Emu
    from oracle_lib import Oracle
    from test_reset_generator import assert_same_record, with_reset_params
    # 100 restarts instead of 1,000: an end-effector target the arm cannot reach runs through ALL of them (robot.py:117-121), and the numpy
    # restatement takes 70 ms per restart
    b = with_reset_params(ModelBlob.load('scratch_itch_' + robot), IK_RESTARTS=100)
    e, o = Emu(b), Oracle(b)
    R = ro.with_collision_check(b.words)
    oks = 0
    for seed, imp in ((1001, -1), (1002, -1), (31, 3), (32, 2)):
        st, info = R.sample(seed, imp)
        se, ie = e.sample(seed, imp)
        assert_same_record(b, st, se, 'seed %d' % seed)
        assert bool(ie[0]) == info['ik_ok'] and int(ie[1]) == info['ik_restarts'] and int(ie[3]) == info['impairment']
        oks += info['ik_ok']
        v = b.view(se[None])
        # ScratchItchEnv.reset (scratch_itch.py:93-132): base on the wheelchair, the arm dynamic, the reactive hold unless the human is an agent
        assert np.allclose(v['base'][0, :3], np.array([0, 0, 0.06]) + b.meta['toc_base'], atol=1e-6) and v['frozen'][0] == 0 and v['total_food'][0] == 1
        if info['impairment'] == 3:
            assert v['human_kp'][0] == 0 and np.any(v['tremor'][0] != 0) and np.all(np.abs(v['tremor'][0]) <= np.deg2rad(10) + 1e-6)
        else:
            assert np.isclose(v['human_kp'][0], 0.01) and np.isclose(v['human_maxf'][0], info['strength'])
        # generate_target (:134-146): a point on the cylinder surface of the drawn limb
        limb = int(v['task'][0, L.SI['LIMB']])
        t = v['task'][0, L.SI['TARGET']:L.SI['TARGET'] + 3].view(np.float32).astype(np.float64)
        dims = b.task_f('SI_LIMB_DIMS', 8)
        length, radius = dims[4 * int(v['gender'][0]) + 2 * limb], dims[4 * int(v['gender'][0]) + 2 * limb + 1]
        assert limb in (0, 1) and np.isclose(np.hypot(t[0], t[1]), radius, atol=1e-6) and -length - 1e-6 <= t[2] <= -radius + 1e-6
        if info['ik_ok']:
            assert not flags_from_oracle(b, o, se) & 1                       # accepted restarts do not touch the human / the wheelchair
            ee, _ = o.ee_pose(se)
            assert np.linalg.norm(ee - info['target_ee']) < 0.011
        # the sampled state steps: emulator vs oracle
        s1, s2 = se.copy(), se.copy()
        a = np.random.RandomState(seed).uniform(-1, 1, 7).astype(np.float32)
        oo, orr, od, oi = o.step(s1, a)
        eo, er, ed, ei, _ = e.step(s2, a)
        assert oi[6] == ei[6] and np.abs(oo[:-1] - eo[:-1]).max() < 1e-4 and abs(orr - er) < 1e-4 * max(1.0, abs(orr))
    assert oks >= 3
    # the limb dimensions are those of the capsules of the compiled human (human_creation.py)
    from assistive_gym_amd.model.human import HumanModel
    hm = HumanModel('male')
    assert np.allclose(b.task_f('SI_LIMB_DIMS', 8)[:4], [hm.dims['upperarm'][1], hm.dims['upperarm'][0], hm.dims['forearm'][1], hm.dims['forearm'][0]])
    assert np.allclose(b.task_f('SI_LIMB_DIMS', 8)[:4], [0.279, 0.043, 0.257, 0.033], atol=1e-6)         # scratch_itch.py:136


def test_witness_point_of_a_parallel_edge_contact_is_what_the_geometry_level_bounds():
    """tests/golden/scratch_sawyer_parallel_edge_case.npz: the state / action of the one GPU parity case of the suite beyond north_star's 1e-3
    AND beyond the step-level conditioning (ScratchItchSawyer, crafted pressed state, session r04f: device tool force 0.98798 N, oracle
    1.01162 N), written by the GPU test and replayed here on the CPU wave emulator.  What it shows: (1) the emulator reproduces the device
    (2e-5 N), so the deviation is float32 arithmetic of the kernel sources and nothing the hardware adds; (2) the oracle's force responds
    LINEARLY to input perturbations here (3.5e-3 N per 1e-6), i.e. no threshold is crossed -- the witness point of the single contact slides
    along an edge of the scratcher that lies parallel to the forearm; (3) the device's deviation is inside the oracle's response at
    conditioning.GEOM_EPS, the level derived from float32's rounding of world-space vertices over a 1 cm feature."""
    import os
    import conditioning as C
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    from oracle_lib import Oracle
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scratch_sawyer_parallel_edge_case.npz'))
    b = ModelBlob.load('scratch_itch_sawyer'); f = b.obs_dim_robot - 1
    o, e = Oracle(b), Emu(b)
    o_obs = o.step(d['start'].copy(), d['action'])[0]
    e_obs = e.step(d['start'].copy(), d['action'])[0]
    assert abs(e_obs[f] - d['dev_obs'][f]) < 1e-4                                   # (1) emulator == device
    dev = abs(float(d['dev_obs'][f]) - float(o_obs[f]))
    assert dev > 1e-3 * max(1.0, abs(o_obs[f]))                                     # the case IS beyond north_star's bound
    s1 = C.ulp_sensitivity(b, o, d['start'], d['action'], trials=6, rel_eps=1e-6)['obs'][f]
    s3 = C.ulp_sensitivity(b, o, d['start'], d['action'], trials=6, rel_eps=3e-6)['obs'][f]
    sg = C.ulp_sensitivity(b, o, d['start'], d['action'], trials=6, rel_eps=C.GEOM_EPS)['obs'][f]
    assert 2.0 < s3 / s1 < 4.5                                                      # (2) linear response: x3 in, ~x3 out
    assert dev <= C.K_GEOM * sg                                                     # (3)
    assert C.within(dev, o_obs[f], lambda: 0.0, floor=C.force_floor(b), step_sens_fn=lambda: s1, geom_sens_fn=lambda: sg)[0]
