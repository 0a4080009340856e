"""ArmManipulationPR2-v1 / ArmManipulationBaxter-v1 (arm_manipulation_envs.py:15-21: robot_arm = 'both' on a two-armed robot -- two tools,
arm_manipulation.py:15-16) without a GPU: blobs against the reference's tables, the oracle's task layer against an independent numpy
restatement of the two-armed branches of arm_manipulation.py, and the arm_manipulation_l kernel variant (32 DoF, two fixed constraints)
on the wave emulator against the oracle.  PARITY UNPINNED vs PyBullet."""
import os

import numpy as np
import pytest

from assistive_gym_amd.model import xform as X


@pytest.fixture(scope='module', params=['baxter', 'pr2'])
def rb(request):
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    from oracle_lib import Oracle
    b = ModelBlob.load('arm_manipulation_' + request.param)
    fall_oracle = Oracle(b.set_param('HUMAN_GRAVITY_Z', -1.0))

    def fall(st, n):
        st = st.copy()
        for i in range(len(st)):
            fall_oracle.settle(st[i], n)
        return st
    return request.param, b, Oracle(b), Emu(b), fall


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset_arm import make_states
    return make_states(blob, n, seed=seed, **kw)


def tool_pose(b, s, t):
    v = b.view(s.reshape(1, -1))
    return v['free'][0, t, :3].astype(np.float64), v['free'][0, t, 3:7].astype(np.float64)      # refpos = 0: base frame = COM frame


def test_model_tables(rb):
    from assistive_gym_amd.model import compiler as L
    name, b, o, e, fall = rb
    RR, LL, TK = L.ROBOT_RIGHT[name], L.ROBOT_BASE[name], L.ARM_MANIPULATION_DUAL[name]
    assert b.task_kind == L.TASK_ARM_MANIPULATION and (b.act_dim, b.obs_dim, b.nfree) == (14, 45, 2)        # arm_manipulation.py:11: 31 + 14
    assert b.task_i('DUP_ACT') == 0 and b.task_i('TOOL2_BODY') == 1
    arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
    assert [b.robot_i(d, 'PB_INDEX') for d in arm_dofs] == RR['arm'] + LL['arm']                           # right arm, then left arm (robot.py:16)
    assert b.nrobot == 2 * (7 + len(LL['grip'])) and b.ndof == b.nrobot + 10
    grip_dofs = [d for d in range(b.nrobot) if b.robot_i(d, 'PB_INDEX') in RR['grip'] + LL['grip']]
    assert np.allclose([b.robot_f(d, 'QT0') for d in grip_dofs], TK['gripper_target'] * 2)
    assert np.isclose(b.robot_f(arm_dofs[0], 'MAXF'), 20.0) and np.isclose(b.free_f(1, 'MASS'), 1.0)
    r = b.meta['ranges']
    assert r['tool_r'][1] - r['tool_r'][0] == 12 and r['tool_l'][1] - r['tool_l'][0] == 12 and r['tool'] == [r['tool_r'][0], r['tool_l'][1]]
    assert {b.collider(c)['link'] for c in range(*r['robot_grip_r'])} <= RR['gripper_collision']            # tool_right does not collide with these
    assert {b.collider(c)['link'] for c in range(*r['robot_grip_l'])} <= LL['gripper_collision']
    assert b.robot_i(b.task_i('EE_LINK'), 'PB_INDEX') <= RR['ee_pb'] < LL['arm'][0] <= b.robot_i(b.task_i('EE2_LINK'), 'PB_INDEX') <= LL['ee_pb']
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (24, 87)


def test_reset_holds_a_tool_in_each_hand(rb):
    name, b, o, e, fall = rb
    st, infos = _states(b, 3, 1001, arm_settler=fall)
    want_q = X.quat_from_rpy(b.meta['ee_rpy'])
    for i in range(3):
        pr, _ = tool_pose(b, st[i], 0)
        pl, _ = tool_pose(b, st[i], 1)
        ee, q = o.ee_pose(st[i])                                                                           # the right end effector
        assert np.linalg.norm(ee - infos[i]['target_ee_pos']) < 0.031 and min(np.linalg.norm(q - want_q), np.linalg.norm(q + want_q)) < 0.031
        assert np.linalg.norm(pr - ee) < 0.35 and np.linalg.norm(pl - pr) > 0.5                             # tool_left hangs from the other hand, ~1 m away (:158-159)
        assert pl[1] > pr[1]                                                                               # left target y = 0.7, right y = -0.3
        assert infos[i]['toc_goals'] >= 3
    obs = o.observe(st[0])
    assert obs.shape == (45,) and not np.allclose(obs[:7], obs[7:14])


def _restated(b, o, s, a, info):
    """the two-armed branch of arm_manipulation.py:24-44 from the oracle's post-step poses and reported preferences"""
    pos, _ = o.fk(s)
    v = b.view(s.reshape(1, -1))
    nr = b.nrobot
    pr, pl = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 1, :3].astype(np.float64)
    elbow, wrist = pos[nr + 7], pos[nr + 9]
    stomach = v['human'][0, b.task_i('STOMACH_BODY'), :3].astype(np.float64)
    waist = v['human'][0, b.task_i('WAIST_BODY'), :3].astype(np.float64)
    rd_human = -np.linalg.norm(elbow - stomach) - np.linalg.norm(wrist - waist)
    return 0.5 * rd_human + 0.25 * -np.linalg.norm(pl - elbow) + 0.25 * -np.linalg.norm(pr - wrist) + 0.01 * -np.linalg.norm(a) + info[5]


def test_oracle_reward_and_action_semantics(rb):
    name, b, o, e, fall = rb
    st, _ = _states(b, 1, 1003, arm_settler=fall)
    s = st[0].copy()
    for k in range(3):
        a = np.random.RandomState(k).uniform(-1, 1, 14).astype(np.float32)
        obs, rew, done, info = o.step(s, a)
        assert abs(_restated(b, o, s, a, info) - rew) < 1e-5
        v = b.view(s.reshape(1, -1))
        arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
        ang = (v['q'][0, arm_dofs].astype(np.float64) + np.pi) % (2 * np.pi) - np.pi
        assert np.abs(obs[14:28] - ang).max() < 1e-5                                                       # 14 joint angles: right arm, left arm
    # the first seven actions drive the right arm, the last seven the left arm
    arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
    s0, s1 = st[0].copy(), st[0].copy()
    o.step(s0, np.concatenate([np.ones(7), np.zeros(7)]).astype(np.float32))
    o.step(s1, np.concatenate([np.zeros(7), np.ones(7)]).astype(np.float32))
    q = b.view(st)['q'][0]
    d0, d1 = np.abs(b.view(s0.reshape(1, -1))['q'][0] - q), np.abs(b.view(s1.reshape(1, -1))['q'][0] - q)
    assert d0[arm_dofs[:7]].max() > 1e-3 and d0[arm_dofs[7:]].max() < 2e-4
    assert d1[arm_dofs[7:]].max() > 1e-3 and d1[arm_dofs[:7]].max() < 2e-4


def lifting_state(b, o, seed, tool, depth=0.003):
    """the right arm of the human stretched out level beyond the mattress (as tests/test_arm_manipulation.py), the highest point of tool
    `tool` `depth` inside the underside of the forearm"""
    st, infos = _states(b, 1, seed)
    s = st[0].copy()
    v = b.view(s.reshape(1, -1))
    nr = b.nrobot
    v['q'][0, nr + 3:nr + 7] = [np.deg2rad(60), 0.0, np.deg2rad(-90), 0.0]
    v['qt'][0, nr:] = v['q'][0, nr:]
    v['tremor_target'][0] = v['q'][0, nr:]
    pos, rot = o.fk(s)
    el, wr = pos[nr + 7], pos[nr + 9]
    g = 'human_male' if infos[0]['gender'] == 'male' else 'human_female'
    rad = [b.collider(k) for k in range(*b.meta['ranges'][g]) if b.collider(k)['link'] == 7][0]['radius']
    fp, fq = tool_pose(b, s, tool)
    hv = np.concatenate([X.apply(fp, fq, b.collider(c)['verts']) for c in range(*b.meta['ranges']['tool_r' if tool == 0 else 'tool_l'])])
    top = hv[np.argmax(hv[:, 2])]
    d = (0.5 * (el + wr) - np.array([0, 0, rad + 0.0025 - depth]) - top).astype(np.float32)
    v['base'][0, :3] += d
    v['free'][0, :2, :3] += d
    # the OTHER arm goes to the middle of its joint ranges (the shift of the base would otherwise drag its tool into the mattress)
    from assistive_gym_amd.host.reset_arm import ArmManipulationSawyerReset
    rs = ArmManipulationSawyerReset(b)
    qs = [np.array([v['q'][0, dd] for dd in a.chain], dtype=np.float64) for a in (rs.arm, rs.arm2)]
    from assistive_gym_amd.model import compiler as L
    other = rs.arm2 if tool == 0 else rs.arm
    name = b.meta['robot']
    tucked = (L.FEEDING_ROBOTS[name] if tool == 0 else L.ROBOT_BASE[name])['frozen_rest']        # reset_joints' tucked pose of that arm (pr2.py:64-65, baxter.py:66-67)
    lo, hi = np.where(other.lower < -1e9, -np.pi, other.lower), np.where(other.upper > 1e9, np.pi, other.upper)
    cands = [np.array([tucked[b.robot_i(dd, 'PB_INDEX')] for dd in other.chain]), 0.5 * (lo + hi)] + list(np.random.RandomState(seed).uniform(lo, hi, size=(40, other.n)))
    other_body = 200 + (1 - tool)
    for cand in cands:
        qs[1 - tool] = cand
        rs._place(v, v['base'][0, :3].astype(np.float64), v['base'][0, 3:].astype(np.float64), qs[0], qs[1])
        # free: neither the other tool nor the arm that carries it touches anything
        if not [r for r in o.collide(s) if r[11] < -0.003 and (other_body in (b.collider(int(r[0]))['body'], b.collider(int(r[1]))['body']) or
                                                               any(b.collider(int(r[k]))['tag'] == 1 and b.collider(int(r[k]))['body'] in other.chain for k in (0, 1)))]:
            break
    else:
        raise AssertionError('no free pose for the other tool')
    return s


@pytest.mark.parametrize('tool', [0, 1])
def test_emulator_matches_oracle_with_a_tool_under_the_forearm(rb, tool):
    name, b, o, e, fall = rb
    if not os.environ.get('AGX_FULL_TESTS') and (name, tool) not in (('baxter', 0), ('pr2', 1)):
        pytest.skip('lean CPU suite: one tool per robot (AGX_FULL_TESTS=1 runs both; the GPU suite covers both robots)')
    s = lifting_state(b, o, 1001, tool)
    so, se = s.copy(), s.copy()
    seen = False
    for k in range(3):
        a = (np.random.RandomState(k).uniform(-1, 1, 14) * 0.1).astype(np.float32)
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6] and oi[7] == ei[7] and oi[4] == ei[4]
        assert np.abs(oo[:43] - eo[:43]).max() < 1e-4 and abs(orr - er) < 1e-4 * max(1.0, abs(orr))
        import conditioning as C
        ff = C.force_floor(b)               # 1e-3 relative, or the float32 floor of a contact force (tests/conditioning.py: the contact spring x 1e-6 m), as every other force comparison
        for c in (43, 44):
            assert abs(oo[c] - eo[c]) <= max(1e-3 * max(1.0, abs(oo[c])), ff), (c, oo[c], eo[c], ff)
        for c in (0, 2, 3):
            assert abs(oi[c] - ei[c]) <= max(1e-3 * max(1.0, abs(oi[c])), ff), (c, oi[c], ei[c], ff)
        assert abs(_restated(b, o, so, a, oi) - orr) < 1e-5
        # [tool_left_force, tool_right_force] (:92): the tool under the forearm is the one that carries force
        if oi[3] > 0:
            seen = True
            assert oo[43 if tool == 1 else 44] >= oi[3] - 1e-4 and oi[0] >= oi[3] - 1e-4 and oi[4] >= 1
    assert seen, 'the forearm rests on the tool'


def test_emulator_free_space_and_coop(rb):
    from emu_lib import Emu
    from oracle_lib import Oracle
    name, b, o, e, fall = rb
    st, _ = _states(b, 1, 1005, arm_settler=fall)
    assert np.abs(o.observe(st[0]) - e.observe(st[0])).max() < 1e-5
    so, se = st[0].copy(), st[0].copy()
    for k in range(2):
        a = np.random.RandomState(k).uniform(-1, 1, 14).astype(np.float32)
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6] and oi[7] == ei[7] and np.abs(oo - eo).max() < 2e-5 and abs(orr - er) < 2e-5
    c = b.coop()
    stc, _ = _states(c, 1, 1005, arm_settler=fall)
    oc, ec = Oracle(c), Emu(c)
    obs = oc.observe(stc[0])
    assert obs.shape == (87,) and not np.allclose(obs[45:52], obs[52:59])      # human_obs: tool_right, tool_left (:100-101)
    so, se = stc[0].copy(), stc[0].copy()
    for k in range(2):
        a = np.random.RandomState(k).uniform(-1, 1, 24).astype(np.float32)
        oo, orr, od, oi = oc.step(so, a)
        eo, er, ed, ei, _ = ec.step(se, a)
        assert oi[6] == ei[6] and np.abs(oo - eo).max() < 2e-5 and abs(orr - er) < 2e-5
