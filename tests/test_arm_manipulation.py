"""ArmManipulationSawyer-v1 (the sixth task, on the single-arm Sawyer) without a GPU: the model blob against the
reference's numbers, the oracle's task layer against an independent numpy restatement of arm_manipulation.py, and the device code
(arm_manipulation kernel variant compiled for the CPU wave emulator) against the oracle.  PARITY UNPINNED vs PyBullet as everywhere."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X


@pytest.fixture(scope='module')
def am():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('arm_manipulation_sawyer')


@pytest.fixture(scope='module')
def am_oracle(am):
    from oracle_lib import Oracle
    return Oracle(am)


@pytest.fixture(scope='module')
def am_emu(am):
    from emu_lib import Emu
    return Emu(am)


@pytest.fixture(scope='module')
def fall(am):
    """the arm's fall at gravity -1 (arm_manipulation.py:125,145-146), on the oracle"""
    from oracle_lib import Oracle
    o = Oracle(am.set_param('HUMAN_GRAVITY_Z', -1.0))

    def run(st, n):
        st = st.copy()
        for i in range(len(st)):
            o.settle(st[i], n)
        return st
    return run


def _states(am, n, seed, **kw):
    from assistive_gym_amd.host.reset_arm import make_states
    return make_states(am, n, seed=seed, **kw)


def tool_hulls_world(am, s):
    v = am.view(s.reshape(1, -1))
    fp, fq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
    return [X.apply(fp, fq, am.collider(c)['verts']) for c in range(*am.meta['ranges']['tool'])]


def scooper_under_forearm(am, o, seed=1001, depth=0.003):
    """the right arm stretched out level beyond the edge of the mattress (no fall), the scooper's highest point `depth` inside the
    underside of the forearm, half way along it"""
    st, infos = _states(am, 1, seed)
    s = st[0].copy()
    v = am.view(s.reshape(1, -1))
    nr = am.nrobot
    v['q'][0, nr + 3:nr + 7] = [np.deg2rad(60), 0.0, np.deg2rad(-90), 0.0]           # shoulder x / y / z, elbow
    v['qt'][0, nr:] = v['q'][0, nr:]
    v['tremor_target'][0] = v['q'][0, nr:]
    pos, rot = o.fk(s)
    el, wr = pos[nr + 7], pos[nr + 9]
    assert abs((wr - el)[2]) < 0.15 * np.linalg.norm(wr - el) and el[0] < -0.55       # level, clear of the mattress
    g = 'human_male' if infos[0]['gender'] == 'male' else 'human_female'
    rad = [am.collider(k) for k in range(*am.meta['ranges'][g]) if am.collider(k)['link'] == 7][0]['radius']
    hv = np.concatenate(tool_hulls_world(am, s))
    top = hv[np.argmax(hv[:, 2])]
    d = 0.5 * (el + wr) - np.array([0, 0, rad + 0.0025 - depth]) - top
    v['base'][0, :3] += d.astype(np.float32)
    v['free'][0, 0, :3] += d.astype(np.float32)
    return s, infos[0]


# ---- model data ------------------------------------------------------------------------------------------------------
def test_model_header_and_tables(am):
    from assistive_gym_amd.model import compiler as L
    assert am.task_kind == L.TASK_ARM_MANIPULATION
    # arm_manipulation.py:11 with robot_arm = 'both' on a single-arm robot (robot.py:16): 31 + 14 observations, 14 actions
    assert (am.ndof, am.nrobot, am.nhdof, am.nfree, am.act_dim, am.obs_dim) == (20, 10, 10, 1, 14, 45)
    c = am.coop()
    assert (c.act_dim, c.obs_dim) == (24, 45 + 32 + 10)
    assert [am.robot_i(d, 'PB_INDEX') for d in range(10)] == [3, 4, 8, 9, 10, 11, 13, 16, 20, 22]
    assert [am.robot_i(d, 'ACT') for d in range(10)] == [7, -1, 8, 9, 10, 11, 12, 13, -1, -1]                  # the second copy drives the motors
    assert am.task_i('DUP_ACT') == 7
    assert np.isclose(am.robot_f(8, 'QT0'), 0.01) and np.isclose(am.robot_f(9, 'QT0'), -0.01)                 # sawyer.py:24
    assert np.isclose(am.robot_f(0, 'KP'), 0.05) and np.isclose(am.robot_f(0, 'MAXF'), 20.0)                  # arm_manipulation.py:114
    assert np.isclose(am.robot_f(12, 'KP'), 0.05) and np.isclose(am.robot_f(12, 'MAXF'), 2.0)                 # :115
    assert np.isclose(am.param('HUMAN_GRAVITY_Z'), -9.81) and am.param('ROBOT_GRAVITY_Z') == 0.0              # :120-121,176
    assert np.isclose(am.free_f(0, 'MASS'), 1.0)                                                             # tool.py:10
    r = am.meta['ranges']
    assert r['tool'][1] - r['tool'][0] == 12                                                                 # the scooper's VHACD pieces
    hv = np.concatenate([am.collider(k)['verts'] for k in range(*r['tool'])])
    assert np.allclose(hv.min(0), [-0.14, 0.0, -0.04], atol=0.005) and np.allclose(hv.max(0), [0.05, 0.10, 0.04], atol=0.005)   # mesh scale 0.001
    assert all(am.collider(k)['friction'] == pytest.approx(0.3) for k in range(*r['bed']))                   # :136
    assert np.isclose(am.task_f('W_DISTANCE'), 0.5) and np.isclose(am.task_f('W_WIPE'), 0.25) and np.isclose(am.task_f('SUCCESS_FRAC'), -0.7)   # config.ini:33-37
    assert np.isclose(am.task_f('C_P'), 0.01) and np.isclose(am.task_f('PRESSURE_DIST'), 0.01)               # config.ini:46, env.py:264
    hb = am.meta['human_bodies']
    assert hb[am.task_i('STOMACH_BODY')] == 24 and hb[am.task_i('WAIST_BODY')] == 27                         # human.py:31-32


def test_reset_sampler_properties(am, am_oracle, fall):
    st, infos = _states(am, 3, 1001, arm_settler=fall)
    v = am.view(st)
    posed, _ = _states(am, 3, 1001)
    vp = am.view(posed)
    for i in range(3):
        assert infos[i]['impairment'] in ('none', 'limits', 'weakness')                                      # human_impairment='no_tremor'
        assert v['frozen'][i] == 0 and np.all(v['tremor'][i] == 0)
        assert np.isclose(v['human_kp'][i], 0.05) and np.isclose(v['human_maxf'][i], 0.01 * infos[i]['strength'])   # :140, human.py:104,126
        # the pose the arm is given before it falls (:139), limits enforced
        assert np.isclose(vp['q'][i, 10 + 6], 0.0, atol=1e-6) and vp['q'][i, 10 + 3] <= np.deg2rad(60) + 1e-6
        assert np.array_equal(v['tremor_target'][i], vp['q'][i, 10:])                                       # target_joint_angles: the posed arm
        assert np.abs(v['q'][i, 10:] - vp['q'][i, 10:]).max() > 0.05                                        # it fell
        assert infos[i]['toc_goals'] >= 1
        pos, _ = am_oracle.fk(st[i])
        assert 0.3 < pos[10 + 9][2] < 1.1                                                                   # the wrist is beside / on the mattress
    o = am_oracle.observe(st[0])
    assert o.shape == (45,) and np.isfinite(o).all() and o[-1] == 0 and o[-2] == 0
    assert np.array_equal(o[:7], o[7:14]) and np.array_equal(o[14:21], o[21:28])                            # the one tool twice, the arm joints twice


# ---- oracle task layer vs an independent numpy restatement of arm_manipulation.py ----------------------------------------
def _restated_reward(am, o, s, a, info, coop=False):
    """reward of arm_manipulation.py:24-44 from the oracle's post-step poses and reported forces"""
    pos, _ = o.fk(s)
    v = am.view(s.reshape(1, -1))
    nr = am.nrobot
    tp = v['free'][0, 0, :3].astype(np.float64)                                       # refpos = 0: the base frame is the COM frame
    elbow, wrist = pos[nr + 7], pos[nr + 9]
    stomach = v['human'][0, am.task_i('STOMACH_BODY'), :3].astype(np.float64)
    waist = v['human'][0, am.task_i('WAIST_BODY'), :3].astype(np.float64)
    rd_left = -np.linalg.norm(tp - elbow)
    rd_human = -np.linalg.norm(elbow - stomach) - np.linalg.norm(wrist - waist)
    return 0.5 * rd_human + 2 * 0.25 * rd_left + 0.01 * -np.linalg.norm(a) + info[5], rd_human


def test_oracle_reward_decomposition_in_free_space(am, am_oracle, fall):
    st, _ = _states(am, 1, 1003, arm_settler=fall)
    s = st[0].copy()
    best = 0.0
    for k in range(4):
        a = np.random.RandomState(k).uniform(-1, 1, 14).astype(np.float32)
        obs, rew, done, info = am_oracle.step(s, a)
        r, rd_human = _restated_reward(am, am_oracle, s, a, info)
        assert abs(r - rew) < 1e-5
        best = rd_human if (best == 0 or rd_human > best) else best
        assert np.isclose(am.view(s.reshape(1, -1))['task'][0, 0:1].view(np.float32)[0], best, atol=1e-6)    # task_success (:47-48)
        assert info[1] == float(best >= -0.7)
        assert info[0] == 0 and info[4] == 0 and info[5] <= 0                       # no contact: preferences = -0.25 * 2 |v_ee|
        assert not done


def test_second_copy_of_the_arm_actions_drives_the_motors(am, am_oracle, fall):
    """setJointMotorControlArray takes the 14 (duplicated) joint indices in order: the targets of the second copy win (robot.py:16)"""
    st, _ = _states(am, 1, 1004, arm_settler=fall)
    arm = [d for d in range(10) if am.robot_i(d, 'ACT') >= 0]
    s0, s1 = st[0].copy(), st[0].copy()
    a0 = np.concatenate([np.ones(7), np.zeros(7)]).astype(np.float32)
    a1 = np.concatenate([np.zeros(7), np.ones(7)]).astype(np.float32)
    _, r0, _, i0 = am_oracle.step(s0, a0)
    _, r1, _, i1 = am_oracle.step(s1, a1)
    q = am.view(st)['q'][0]
    assert np.abs(am.view(s0.reshape(1, -1))['q'][0, arm] - q[arm]).max() < 1e-4          # first copy: overwritten
    assert np.abs(am.view(s1.reshape(1, -1))['q'][0, arm] - q[arm]).max() > 1e-3          # second copy: moves the arm
    assert abs((r0 - i0[5]) - (r1 - i1[5])) < 0.05                                         # both pay the same action penalty


def test_oracle_lifting_contact_and_pressure_term(am, am_oracle):
    s, info0 = scooper_under_forearm(am, am_oracle)
    seen = False
    for k in range(4):
        a = np.zeros(14, dtype=np.float32)
        obs, rew, done, info = am_oracle.step(s, a)
        total_f, robot_f, th_f, near, pref = info[0], info[2], info[3], info[4], info[5]
        assert np.isclose(total_f, robot_f + 2 * th_f, rtol=1e-6, atol=1e-6)          # the one tool counted as right and left (:68)
        assert obs[-1] == obs[-2] and obs[-1] >= th_f - 1e-6                          # tool_force: every contact of the tool
        if th_f > 0:
            seen = True
            assert near >= 1                                                        # a touching pair is within 1 cm (env.py:264)
            # preferences = -0.25 * 2|v| - 0.01 * robot_f - 0.01 * 2 * th_f / near  (env.py:266-274)
            v_term = pref + 0.01 * robot_f + 0.01 * 2 * th_f / near
            assert v_term <= 1e-6
        r, _ = _restated_reward(am, am_oracle, s, a, info)
        assert abs(r - rew) < 1e-5
    assert seen, 'the arm came to rest on the scooper'


# ---- device code on the wave emulator vs the oracle ----------------------------------------------------------------------
def _compare(am, o, e, s, actions, tol=2e-5):
    so, se = s.copy(), s.copy()
    for a in actions:
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6] and oi[7] == ei[7], 'same contacts, same rows'
        assert np.abs(oo - eo).max() < tol * max(1.0, np.abs(oo).max()) and abs(orr - er) < tol * max(1.0, abs(orr)) and od == ed
        assert oi[4] == ei[4] and oi[1] == ei[1]                              # near pairs, task success flag
        for k in (0, 2, 3):
            assert abs(oi[k] - ei[k]) <= 1e-3 * max(1.0, abs(oi[k]))          # forces: 1e-3 relative (north star)
        vo, ve = am.view(so.reshape(1, -1)), am.view(se.reshape(1, -1))
        assert np.abs(vo['q'] - ve['q']).max() < tol and np.abs(vo['task'].view(np.float32) - ve['task'].view(np.float32)).max() < tol
    return so, se


def test_emulator_free_space(am, am_oracle, am_emu, fall):
    st, _ = _states(am, 2, 1001, arm_settler=fall)
    for i in range(2):
        _compare(am, am_oracle, am_emu, st[i], [np.random.RandomState(10 * i + k).uniform(-1, 1, 14).astype(np.float32) for k in range(3)])


def test_emulator_lifting_contact(am, am_oracle, am_emu):
    s, _ = scooper_under_forearm(am, am_oracle)
    so, se = _compare(am, am_oracle, am_emu, s, [np.random.RandomState(k).uniform(-1, 1, 14).astype(np.float32) * 0.1 for k in range(3)], tol=1e-4)


def test_emulator_observe_and_coop(am, am_oracle, am_emu, fall):
    from oracle_lib import Oracle
    from emu_lib import Emu
    st, _ = _states(am, 1, 1005, arm_settler=fall)
    assert np.abs(am_oracle.observe(st[0]) - am_emu.observe(st[0])).max() < 1e-5
    c = am.coop()
    stc, _ = _states(c, 1, 1005, arm_settler=fall)
    assert np.all(c.view(stc)['human_kp'] == 0)                               # the human's motors follow its actions (env.py:190-219)
    oc, ec = Oracle(c), Emu(c)
    obs = oc.observe(stc[0])
    assert obs.shape == (87,) and np.array_equal(obs[45:52], obs[52:59])      # human_obs: the one tool twice (:100-101)
    so, se = _compare(c, oc, ec, stc[0], [np.random.RandomState(k).uniform(-1, 1, 24).astype(np.float32) for k in range(3)])
    q0 = c.view(stc)['q'][0, 10:]
    assert np.abs(c.view(so.reshape(1, -1))['q'][0, 10:] - q0).max() > 1e-3   # the human's actions move its arm
