"""ScratchItchPR2-v1 / ScratchItchPR2Human-v1 (BASELINE config 4's environment) without a GPU: model blob against the reference's
numbers, the oracle's task layer against bed_bathing-independent numpy restatements of scratch_itch.py, and the device code
(scratch_itch kernel variant on the CPU wave emulator) against the oracle.  PARITY UNPINNED vs PyBullet (oracle/agx_oracle.h)."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X


@pytest.fixture(scope='module')
def si():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('scratch_itch_pr2')


@pytest.fixture(scope='module')
def si_oracle(si):
    from oracle_lib import Oracle
    return Oracle(si)


@pytest.fixture(scope='module')
def si_emu(si):
    from emu_lib import Emu
    return Emu(si)


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset_scratch import make_states
    return make_states(blob, n, seed=seed, **kw)


def tip_pose(blob, state):
    """world pose of tool link 1 (the scratcher's tip)"""
    v = blob.view(state.reshape(1, -1))
    fp, fq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
    bp, bq = X.compose(fp, fq, blob.free_f(0, 'REFPOS', 3), blob.free_f(0, 'REFQUAT', 4))
    return X.compose(bp, bq, blob.task_f('TOOL_OBS_POS', 3), blob.task_f('TOOL_OBS_QUAT', 4))


def target_world(blob, oracle, state):
    v = blob.view(state.reshape(1, -1))
    pos, rot = oracle.fk(state)
    link = blob.task_i_n('ARM_LINK', 2)[int(v['task'][0, 3])]
    return rot[link] @ v['task'][0, 0:3].view(np.float32).astype(np.float64) + pos[link], pos[link], rot[link]


def scratching_state(blob, oracle, seed=1001, depth=0.003, **kw):
    """a post-reset state with the robot base translated so that the scratcher's tip (sphere r = 1 cm) presses `depth` into the skin at the target"""
    st, infos = _states(blob, 1, seed, **kw)
    s = st[0].copy()
    tgt, lp, lR = target_world(blob, oracle, s)
    axis = lR @ np.array([0, 0, -1.0])
    radial = (tgt - lp) - np.dot(tgt - lp, axis) * axis
    radial /= np.linalg.norm(radial)
    want = tgt + radial * (0.01 - depth)
    p, _ = tip_pose(blob, s)
    v = blob.view(s.reshape(1, -1))
    d = (want - p).astype(np.float32)
    v['base'][0, :3] += d; v['free'][0, 0, :3] += d; v['free'][0, 0, 7:] = 0
    return s


def test_model_header_and_tables(si):
    from assistive_gym_amd.model import compiler as L
    assert si.task_kind == L.TASK_SCRATCH_ITCH
    assert (si.ndof, si.nrobot, si.nhdof, si.nfree, si.act_dim, si.obs_dim) == (21, 11, 10, 1, 7, 30)       # scratch_itch.py:8: 23 + 7
    assert [si.robot_i(d, 'PB_INDEX') for d in range(11)] == [64, 65, 66, 68, 69, 71, 72, 79, 80, 81, 82]   # pr2.py:9,14
    assert [si.robot_i(d, 'ACT') for d in range(11)] == [0, 1, 2, 3, 4, 5, 6, -1, -1, -1, -1]
    assert all(np.isclose(si.robot_f(d, 'QT0'), 0.25) and np.isclose(si.robot_f(d, 'MAXF'), 500.0) for d in range(7, 11))   # pr2.py:17, robot.py:76-79
    assert [si.robot_i(d, 'HAS_LIMIT') for d in range(7)] == [1, 1, 1, 1, 0, 1, 0]                          # forearm / wrist roll are continuous
    assert np.isclose(si.robot_f(0, 'MASS'), 25.799322, rtol=1e-6)                                           # URDF_USE_INERTIA_FROM_FILE masses (pr2.py:52)
    assert np.isclose(si.robot_f(0, 'JDAMP'), 10.0) and np.isclose(si.robot_f(4, 'JDAMP'), 0.1)
    coop = si.coop()
    assert (coop.act_dim, coop.obs_dim) == (17, 30 + 34) and coop.task_i('ARM_LIMIT_ON') == 1              # scratch_itch.py:8, human.py:136-137
    r = si.meta['ranges']
    assert [si.collider(c)['link'] for c in range(*r['tool'])] == [-1, 0, 1] and si.task_i('PAD_LINK') == 0b110    # linkA in [0, 1]
    assert np.allclose(si.task_f('TOOL_OBS_POS', 3), [0.075, 0, 0]) and np.isclose(si.free_f(0, 'MASS'), 0.3)
    assert np.isclose(si.task_f('SUCCESS_FRAC'), 25.0) and np.isclose(si.task_f('W_WIPE'), 1.0)            # config.ini:3-7
    assert si.param('HUMAN_GRAVITY_Z') == 0.0 and si.param('ROBOT_GRAVITY_Z') == 0.0                        # scratch_itch.py:121-124
    # the static branches of the PR2 (base, torso, head, tucked right arm, lasers) are world-fixed geometry of the robot body
    assert r['robot_base'][1] - r['robot_base'][0] >= 25


def test_reset_sampler_properties(si, si_oracle):
    st, infos = _states(si, 6, 3001)
    v = si.view(st)
    for i in range(6):
        info = infos[i]
        tgt, lp, lR = target_world(si, si_oracle, st[i])
        # the target lies on the surface of its limb's capsule (Util.point_on_capsule)
        from assistive_gym_amd.model.human import HumanModel
        radius, length = HumanModel(info['gender']).dims['upperarm' if info['limb'] == 0 else 'forearm']
        loc = lR.T @ (tgt - lp)
        assert np.isclose(np.hypot(loc[0], loc[1]), radius, atol=1e-5) and -length <= loc[2] <= -radius + 1e-6
        assert v['frozen'][i] == 0                                             # reactive_force = 1: the arm stays dynamic (human.py:108)
        agent = info['impairment'] == 'tremor'
        assert (v['human_kp'][i] == 0) == agent and (agent or np.isclose(v['human_maxf'][i], info['strength']))
        p, _ = tip_pose(si, st[i])
        assert np.linalg.norm(p - info['target_ee_pos']) < 0.3 and v['total_food'][i] == 1
    o = si_oracle.observe(st[0])
    assert o.shape == (30,) and np.isfinite(o).all() and o[-1] == 0
    assert np.allclose(o[7:10], o[0:3] - o[10:13], atol=1e-6)                  # tool_pos_real - target_pos_real, target_pos_real


def test_oracle_scratch_logic(si, si_oracle):
    """reward = -|target - tip| - 0.01 |a| + 5 [scratch] + preferences (scratch_itch.py:14-33); a scratch needs the contact point to have
    moved > 1 cm since the last rewarded one and less than 10 N at the target"""
    s = scratching_state(si, si_oracle)
    v = si.view(s.reshape(1, -1))
    a = np.zeros(7, dtype=np.float32)
    obs, rew, done, info = si_oracle.step(s, a)
    tgt, _, _ = target_world(si, si_oracle, s)
    tip, _ = tip_pose(si, s)
    assert info[4] == 5.0 and v['task_success'][0] == 1                       # prev_target_contact_pos starts at (0, 0, 0): the first contact counts
    prev = v['task'][0, 12:15].view(np.float32).astype(np.float64)
    assert np.linalg.norm(prev - tgt) < 0.025
    assert abs(rew - (-np.linalg.norm(tgt - tip) + 5.0 + info[5])) < 1e-5
    # same place again: the contact point has not moved by 1 cm -> no second reward
    obs, rew2, done, info2 = si_oracle.step(s, a)
    assert info2[4] == 0.0 and v['task_success'][0] == 1
    assert info2[3] <= info2[0] + 1e-6                                        # tool_force_at_target is part of total_force_on_human
    # preferences: -0.25 v_ee - 0.01 (total - at_target) - 0.05 [at_target >= 10] at_target
    hf = 0.0 if info2[3] < 10 else -info2[3]
    assert -(info2[5] + 0.01 * (info2[0] - info2[3]) - 0.05 * hf) / 0.25 >= -1e-6


def _compare(blob, o, e, s, actions, tol=2e-5):
    so, se = s.copy(), s.copy()
    for a in actions:
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6] and abs(oi[7] - ei[7]) <= 4, 'same contacts; limit rows of the fingers held exactly 0.25 rad from their limit may flip'
        f = blob.obs_dim_robot - 1
        dev = np.abs(oo - eo)
        assert dev[f] <= 1e-3 * max(1.0, abs(oo[f]))
        dev[f] = 0
        if blob.is_coop:
            dev[-2:] = np.minimum(dev[-2:], 0) if np.all(dev[-2:] <= 1e-3 * np.maximum(1.0, np.abs(oo[-2:]))) else dev[-2:]
        assert dev.max() < tol and abs(orr - er) < tol * max(1.0, abs(orr)) and od == ed
        assert oi[4] == ei[4] and oi[1] == ei[1]
        for k in (0, 2, 3):
            assert abs(oi[k] - ei[k]) <= 1e-3 * max(1.0, abs(oi[k]))
        vo, ve = blob.view(so.reshape(1, -1)), blob.view(se.reshape(1, -1))
        assert np.abs(vo['q'] - ve['q']).max() < tol and vo['task_success'][0] == ve['task_success'][0]
        assert np.abs(vo['task'][0, 12:15].view(np.float32) - ve['task'][0, 12:15].view(np.float32)).max() < 1e-4
    return so, se


def test_emulator_free_space_and_reactive_hold(si, si_oracle, si_emu):
    st, infos = _states(si, 2, 1001)
    for i in range(2):
        _compare(si, si_oracle, si_emu, st[i], [np.random.RandomState(10 * i + k).uniform(-1, 1, 7).astype(np.float32) for k in range(3)])


def test_emulator_scratching_contact(si, si_oracle, si_emu):
    s = scratching_state(si, si_oracle, depth=0.004)
    so, se = _compare(si, si_oracle, si_emu, s, [np.random.RandomState(k).uniform(-1, 1, 7).astype(np.float32) * 0.1 for k in range(3)], tol=2e-4)
    assert si.view(so.reshape(1, -1))['task_success'][0] >= 1


def test_emulator_coop_with_arm_limits(si):
    from emu_lib import Emu
    from oracle_lib import Oracle
    coop = si.coop()
    o, e = Oracle(coop), Emu(coop)
    st, infos = _states(coop, 1, 7101)
    assert coop.view(st)['human_kp'][0] == 0                                  # a controllable human is an agent: motor_gains 0.05 every step
    so, se = st[0].copy(), st[0].copy()
    for k in range(3):
        a = np.random.RandomState(50 + k).uniform(-1, 1, 17).astype(np.float32)
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oo.shape == (64,) and np.abs(oo - eo).max() < 2e-5 and abs(orr - er) < 2e-5
    vo = coop.view(so.reshape(1, -1))
    assert vo['task'][0, 10] == 1                                             # the classifier accepted (and remembered) the pose
    assert np.abs(oo[30 + 13:30 + 23] - vo['q'][0, coop.nrobot:]).max() < 1e-6   # human part: tool pose (7), tool - target (3), target (3), 10 joint angles, ...
