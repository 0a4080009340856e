"""The Stretch (agents/stretch.py) without a GPU -- FeedingStretch-v1 (feeding_envs.py:33-35) in detail, ScratchItchStretch-v1 and
BedBathingStretch-v1 (scratch_itch_envs.py:31-33, bed_bathing_envs.py:31-33) on the same machinery: the mobile manipulator's model blob
against the reference's robot table, the mobile branch of init_robot_pose on the host (env.py:282-293), the wheels against analytic rolling /
turning cases in the oracle, and the *_m kernel variants on the wave emulator against the oracle.  ArmManipulationStretch is not built: the
reference's own take_step raises for it (Stretch('wheel_both') has 8 controllable joints, its action_multiplier 5 entries: env.py:197).
PARITY UNPINNED vs PyBullet (the physics half)."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X
from conftest import full


@pytest.fixture(scope='module')
def rb():
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    b = ModelBlob.load('feeding_stretch')
    return b, Oracle(b)


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset import make_states
    return make_states(blob, n, seed=seed, **kw)


def test_model_tables(rb):
    from assistive_gym_amd.model import compiler as L
    b, o = rb
    assert b.task_kind == L.TASK_FEEDING and (b.act_dim, b.obs_dim, b.nhdof, b.nfood) == (5, 21, 4, 8)          # feeding.py:10 with 5 joints, 2 of them wheels
    assert b.h['BASE_LINK'] == 6 and b.nrobot == 16 and b.meta['mount'] == 'mobile'
    # six virtual joints (x, y, z, yaw, pitch, roll) in front of the URDF's: massless but the last, which is the base link
    for k in range(6):
        assert b.robot_i(k, 'PB_INDEX') == -1 and b.robot_i(k, 'PARENT') == k - 1 and b.robot_i(k, 'JTYPE') == (1 if k < 3 else 0) and b.robot_i(k, 'ACT') == -1
        assert b.robot_f(k, 'MAXF') == 0 and (b.robot_f(k, 'MASS') > 10.0) == (k == 5)
    pb = [b.robot_i(d, 'PB_INDEX') for d in range(6, 16)]
    assert pb == [0, 1, 3, 5, 6, 7, 8, 9, 11, 13]                                                                # stretch.py:53 + the fingers (:14); head pan / tilt are static
    act = {b.robot_i(d, 'PB_INDEX'): b.robot_i(d, 'ACT') for d in range(6, 16)}
    assert act == {0: 0, 1: 1, 3: 2, 5: 3, 6: 3, 7: 3, 8: 3, 9: 4, 11: -1, 13: -1}                              # action_duplication [1, 1, 1, 4, 1], stretch.py:51
    src = {b.robot_i(d, 'PB_INDEX'): b.robot_i(d, 'ACT_SRC') for d in range(6, 16)}
    d5 = pb.index(5) + 6
    assert src == {0: 0, 1: 0, 3: 0, 5: 0, 6: d5 + 1, 7: d5 + 1, 8: d5 + 1, 9: 0, 11: 0, 13: 0}
    mult = {b.robot_i(d, 'PB_INDEX'): b.robot_f(d, 'ACT_MULT') for d in range(6, 14)}
    assert mult == {0: 3, 1: 3, 3: 2, 5: 1, 6: 1, 7: 1, 8: 1, 9: 2}                                              # stretch.py:52
    gains = [round(b.robot_f(d, 'KP'), 6) for d in range(6, 14)], [b.robot_f(d, 'MAXF') for d in range(6, 14)]
    assert gains == ([0.1] * 2 + [0.01] + [0.025] * 5, [10] * 2 + [20] + [10] * 5)                              # stretch.py:49-50
    assert [b.robot_i(d, 'OBS_SKIP') for d in range(6, 14)] == [1, 1, 0, 0, 1, 1, 1, 0]                          # feeding.py:90-92; the duplicates are no controllable joints
    assert [b.robot_f(d, 'MASS') for d in (6, 7)] == [10, 10]                                                    # stretch.py:82-83
    assert np.isclose(sum(b.robot_f(d, 'MASS') for d in range(8, 16)), 0.1 * 16)                                 # stretch.py:79-80: each of the 16 massive links from the lift on weighs 0.1
    assert b.param('ROBOT_GRAVITY_Z') == np.float32(-9.81)                                                       # feeding.py:150-151
    # the base link touches the ground without friction (stretch.py:86); the wheels keep the URDF default
    C = L.C
    cols = [(b.i[b.h['OFF_COLL'] + k * C['STRIDE'] + C['BODY']], b.i[b.h['OFF_COLL'] + k * C['STRIDE'] + C['LINK']], float(b.f[b.h['OFF_COLL'] + k * C['STRIDE'] + C['FRICTION']]))
            for k in range(b.h['NCOLL'])]
    assert {f for body, link, f in cols if body == 5 and link == -1} == {0.0} and {f for body, link, f in cols if body in (6, 7)} == {0.5}
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (9, 21 + 23)


def test_host_reset_draws_the_mobile_branch(rb):
    b, o = rb
    n = 12
    st, infos = _states(b, n, 4001)
    for i in range(n):
        v = b.view(st[i:i + 1])
        d = v['base'][0, :3] - np.array([-0.9, -0.3, 0.09])                                                     # stretch.py:37
        assert np.all(np.abs(d[:2]) <= 0.1 + 1e-6) and abs(d[2]) < 1e-6                                          # env.py:285-286
        yaw = 2 * np.arctan2(v['base'][0, 5], v['base'][0, 6])
        assert abs(v['base'][0, 3]) < 1e-6 and abs(v['base'][0, 4]) < 1e-6 and abs(yaw - np.pi / 2) <= np.deg2rad(30) + 1e-6      # env.py:287-289
        q = v['q'][0]
        assert np.all(q[:8] == 0) and abs(q[8] - 0.75) <= 0.1 + 1e-6 and np.all(q[9:16] == 0)                    # stretch.py:58-62, gripper feeding: [0, 0]
        spoon, food = v['free'][0, 0, :3], v['free'][0, 2:, :3]
        assert np.all(np.linalg.norm(food - spoon, axis=1) < 0.03)
        # the spoon sits at the tool joint (link 15) offset by tool_pos_offset (stretch.py:27), i.e. near the gripper, 0.6 .. 1.0 m above the ground
        assert 0.6 < spoon[2] < 1.1


@pytest.mark.parametrize('friction', [0.5, 0.1])
def test_wheels_roll_and_turn_as_a_differential_drive(rb, friction):
    """The friction model is one row along the slip direction; a driven wheel must then roll: over two seconds of equal wheel
    actions the base advances by wheel radius x wheel angle minus a few percent of slip, straight; opposite actions turn it on the spot by
    radius x (angle difference) / track; the base stays level on the ground."""
    b, o = rb
    st, _ = _states(b, 1, 4005, impairment='none')
    r, track = 0.0508, 2 * 0.15765                                                                              # stretch_uncalibrated.urdf: wheel mesh radius, joint offsets
    slip = 0.09 if friction >= 0.5 else 0.3
    for name, aw in (('forward', (1, 1)), ('spin', (1, -1)), ('arc', (1, 0.5))):
        s = st[0].copy()
        b.view(s[None])['plane_friction'][0] = friction
        o.settle(s, 25)
        q0 = b.view(s[None])['q'][0].copy()
        assert abs(q0[2]) < 4e-3                                                                                 # placed with its centre of mass at z = 0.09 (stretch.py:37: PyBullet's base frame is the inertial frame), i.e. ON its wheels: it stays there
        a = np.zeros(5, np.float32)
        a[0], a[1] = aw
        for k in range(20):
            obs, rew, done, info = o.step(s, a)
        q1 = b.view(s[None])['q'][0].copy()
        dth = q1[6:8] - q0[6:8]
        # the target of an env step is the wheel angle + 5 x 0.05 x 3 (env.py:188,197,201-216); the position motor with gain 0.1 (stretch.py:49)
        # closes a tenth of the remaining error per substep, well below its 10 N
        assert np.allclose(np.abs(dth), 20 * 0.75 * (1 - 0.9 ** 5) * np.abs(aw), rtol=0.02)
        # q[:3] is the base's centre of mass (the model's base frame, PyBullet's); the point the drive turns about is the middle of the axle =
        # the origin of base_link, at -inertial origin (stretch_uncalibrated.urdf: -0.1095, -0.0007) in that frame
        axle = lambda q: q[:2] + np.array([[np.cos(q[3]), -np.sin(q[3])], [np.sin(q[3]), np.cos(q[3])]]) @ np.array([0.109461304328163, 0.000741018909047708])
        dist, yaw = np.linalg.norm(axle(q1) - axle(q0)), q1[3] - q0[3]
        want_dist, want_yaw = r * dth.mean(), r * (dth[0] - dth[1]) / track
        assert abs(q1[2] - q0[2]) < 1e-3 and np.all(np.abs(q1[4:6]) < 5e-3), name
        if name == 'forward':
            assert (1 - slip) * want_dist <= dist <= 1.01 * want_dist and abs(yaw) < 0.03 and abs(q1[1] - q0[1]) < 0.01, (name, dist, want_dist, yaw)
        elif name == 'spin':
            assert (1 - slip) * want_yaw <= yaw <= 1.01 * want_yaw and dist < 0.02, (name, yaw, want_yaw, dist)
        else:
            assert (1 - slip) * want_yaw <= yaw <= 1.01 * want_yaw and (1 - slip) * want_dist <= dist <= 1.01 * want_dist, (name, yaw, want_yaw, dist, want_dist)


def test_observation_is_taken_in_the_moving_base_frame(rb):
    """convert_to_realworld uses the robot's CURRENT base pose (agent.py:60-64, 142-150): while the robot drives forward with its arm still,
    the spoon's position in the observation stays put and the head's recedes by the distance driven"""
    b, o = rb
    st, _ = _states(b, 1, 4006, impairment='none')
    s = st[0].copy()
    b.view(s[None])['plane_friction'][0] = 0.5
    o.settle(s, 25)
    a = np.zeros(5, np.float32)
    obs0 = o.step(s, a)[0].copy()
    q0 = b.view(s[None])['q'][0].copy()
    a[0] = a[1] = 1.0
    for k in range(10):
        obs1 = o.step(s, a)[0]
    q1 = b.view(s[None])['q'][0].copy()
    moved = np.linalg.norm(q1[:2] - q0[:2])
    assert moved > 0.1
    assert np.abs(obs1[:3] - obs0[:3]).max() < 0.02                           # spoon in the base frame
    assert abs(np.linalg.norm(obs1[13:16] - obs0[13:16]) - moved) < 0.02      # head position in the base frame (feeding.py:99: 7 + 3 + 3 joint angles, then the head)
    assert np.allclose(obs1[10:13], q1[[8, 9, 13]], atol=1e-6)                # lift, arm, wrist: the wheel angles are left out (feeding.py:90-92)


def test_duplicated_action_drives_the_four_telescoping_joints(rb):
    b, o = rb
    st, _ = _states(b, 1, 4007, impairment='none')
    s = st[0].copy()
    o.settle(s, 25)
    a = np.zeros(5, np.float32)
    a[3] = 1.0
    v = b.view(s[None])
    q_before = v['q'][0, 9]
    o.step(s, a)
    v = b.view(s[None])
    # one action, one target for the four joints: joint 5's angle + 5 x 0.05 x 1, clipped at its limit (env.py:201-220)
    want = min(q_before + 5 * 0.05, 0.13)
    assert np.allclose(v['qt'][0, 9:13], want, atol=1e-6)
    for k in range(10):
        o.step(s, a)
    v = b.view(s[None])
    assert np.all(v['q'][0, 9:13] > 0.05) and np.all(v['q'][0, 9:13] <= 0.13 + 1e-4)      # gain 0.025: 2.5 % of the remaining error per substep


@pytest.mark.parametrize('coop', [False, pytest.param(True, marks=full)])
def test_emulator_settle_and_step_match_the_oracle(rb, coop):
    from emu_lib import Emu
    from oracle_lib import Oracle
    b, _ = rb
    b12 = (b.coop() if coop else b).set_param('NITER', 12)
    o12, e12 = Oracle(b12), Emu(b12)
    st, _ = _states(b12, 2, 4010)
    rng = np.random.RandomState(3)
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        o12.settle(so, 8); e12.settle(se, 8)
        assert np.abs(b.view(so[None])['q'] - b.view(se[None])['q']).max() < 1e-5
        s = so
        for k in range(3):
            a = rng.uniform(-1, 1, b12.act_dim).astype(np.float32)
            s1, s2 = s.copy(), s.copy()
            o_obs, o_rew, o_done, o_info = o12.step(s1, a)
            e_obs, e_rew, e_done, e_info, _ = e12.step(s2, a)
            assert o_info[6] == e_info[6] and o_info[7] == e_info[7]
            assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4
            assert np.abs(b.view(s1[None])['q'] - b.view(s2[None])['q']).max() < 5e-5
            s = s1


# ---- the other tasks of the Stretch ----------------------------------------------------------------------------------------------------
def _task_states(blob, n, seed, **kw):
    from assistive_gym_amd.host import reset_bed, reset_scratch
    from assistive_gym_amd.model import compiler as L
    if blob.task_kind == L.TASK_SCRATCH_ITCH:
        return reset_scratch.make_states(blob, n, seed=seed, **kw)[0]
    return reset_bed.make_states(blob, n, seed=seed, **kw)[0]


@pytest.fixture(scope='module', params=['scratch_itch', 'bed_bathing'])
def tb(request):
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    b = ModelBlob.load(request.param + '_stretch')
    return request.param, b, Oracle(b)


def test_other_tasks_model_tables(tb):
    from assistive_gym_amd.model import compiler as L
    task, b, o = tb
    base_obs = {'scratch_itch': 23, 'bed_bathing': 17}[task]                                   # scratch_itch.py:8, bed_bathing.py:10
    assert (b.act_dim, b.obs_dim, b.nrobot, b.nhdof) == (5, base_obs + 3, 16, 10) and b.h['BASE_LINK'] == 6
    assert b.meta['mount'] == 'mobile' and b.has_reset_generator                                  # (bed bathing: with the rag-doll model attached)
    assert b.meta['mobile_base'] == {'scratch_itch': [-1.0, -0.1, 0.09], 'bed_bathing': [-1.1, -0.1, 0.09]}[task]        # stretch.py:37,39
    assert b.meta['lift'] == {'scratch_itch': 0.75, 'bed_bathing': 0.95}[task]                                             # stretch.py:58-62
    assert [b.robot_f(d, 'QT0') for d in (14, 15)] == [np.float32(0.1)] * 2                                                # gripper_pos, stretch.py:21,24
    assert b.param('ROBOT_GRAVITY_Z') == np.float32(-9.81) and b.param('HUMAN_GRAVITY_Z') == (0.0 if task == 'scratch_itch' else -1.0)
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (15, 2 * base_obs + 3 + 1 + 10)


def test_other_tasks_reset_stand_and_drive(tb):
    task, b, o = tb
    st = _task_states(b, 3, 4101)
    for i in range(3):
        s = st[i].copy()
        v = b.view(s[None])
        assert np.all(np.abs(v['base'][0, :2] - np.array(b.meta['mobile_base'][:2])) <= 0.1 + 1e-6) and v['base'][0, 2] == np.float32(0.09)
        assert abs(v['q'][0, 8] - b.meta['lift']) <= 0.1 + 1e-6 and np.all(v['q'][0, :8] == 0)
        v['plane_friction'][0] = 0.5
        o.settle(s, 15)
        q0 = b.view(s[None])['q'][0].copy()
        assert abs(q0[2]) < 4e-3 and np.all(np.abs(q0[3:6]) < 0.02)
        a = np.zeros(b.act_dim, np.float32)
        a[0] = a[1] = 1.0
        for k in range(10):
            obs, rew, done, info = o.step(s, a)
        q1 = b.view(s[None])['q'][0].copy()
        dist, want = np.linalg.norm(q1[:2] - q0[:2]), 0.0508 * (q1[6:8] - q0[6:8]).mean()
        assert 0.9 * want <= dist <= 1.01 * want and np.isfinite(obs).all() and np.isfinite(rew)


@pytest.mark.parametrize('coop', [False, pytest.param(True, marks=full)])
def test_other_tasks_emulator_matches_the_oracle(tb, coop):
    from emu_lib import Emu
    from oracle_lib import Oracle
    task, b, _ = tb
    b12 = (b.coop() if coop else b).set_param('NITER', 12)
    o12, e12 = Oracle(b12), Emu(b12)
    st = _task_states(b12, 1, 4110)
    rng = np.random.RandomState(5)
    s = st[0].copy()
    o12.settle(s, 8)                      # onto the ground
    for k in range(3):
        a = rng.uniform(-1, 1, b12.act_dim).astype(np.float32)
        s1, s2 = s.copy(), s.copy()
        o_obs, o_rew, o_done, o_info = o12.step(s1, a)
        e_obs, e_rew, e_done, e_info, _ = e12.step(s2, a)
        assert o_info[6] == e_info[6] and o_info[7] == e_info[7]
        assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4
        assert np.abs(b.view(s1[None])['q'] - b.view(s2[None])['q']).max() < 5e-5
        s = s1


# ---- DressingStretch-v1 (dressing_envs.py:31-33): the rigid scene + the garment hanging from the gripper ------------------------------------
@pytest.fixture(scope='module')
def db():
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    b = ModelBlob.load('dressing_stretch')
    return b, Oracle(b)


def test_dressing_model_tables(db):
    from assistive_gym_amd.model import compiler as L
    b, o = db
    assert b.task_kind == L.TASK_DRESSING and (b.act_dim, b.obs_dim, b.nrobot, b.nhdof, b.nfree) == (5, 17 + 3, 16, 10, 0) and b.h['SIM_SUBSTEPS'] == 8      # dressing.py:9,184
    assert b.h['BASE_LINK'] == 6 and b.meta['mount'] == 'mobile' and b.has_reset_generator and b.meta['mobile_yaw'] is False
    assert b.meta['mobile_base'] == [0.75, -0.4, 0.09] and np.allclose(b.meta['mobile_rpy'], [0, 0, -np.pi / 2]) and b.meta['lift'] == 0.95                  # stretch.py:41,47,59-60
    # the motor gains are divided by numSubSteps (dressing.py:135-137)
    assert np.allclose([b.robot_f(d, 'KP') for d in range(6, 14)], np.array([0.1] * 2 + [0.01] + [0.025] * 5) / 8.0)
    assert [b.robot_f(d, 'MAXF') for d in range(6, 14)] == [10] * 2 + [20] + [10] * 5
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (15, 20 + 28)


def test_dressing_reset_garment_and_emulator_parity(db):
    from emu_lib import Emu
    from assistive_gym_amd.host.reset_dressing import make_states
    from assistive_gym_amd.model import compiler as L
    from test_dressing import cloth_tables
    b, o = db
    e = Emu(b)
    st, cloth, infos = make_states(b, 2, seed=47)
    t = cloth_tables(b)
    for i in range(2):
        v = b.view(st[i:i + 1])
        assert np.all(np.abs(v['base'][0, :2] - np.array([0.75, -0.4])) <= 0.1 + 1e-6) and np.allclose(v['base'][0, 3:], X.quat_from_rpy([0, 0, -np.pi / 2]), atol=1e-6)   # no yaw draw
        ee, q = o.ee_pose(st[i])
        assert np.allclose(ee, infos[i]['start_ee_pos'], atol=1e-5)
        assert np.allclose(cloth[i, 0] - t['x0'], ee - np.array(b.meta['cloth_orig_pos']), atol=1e-5)            # dressing.py:148-153
    s, c = st[0].copy(), cloth[0].copy()
    b.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = np.array([-9.81 / 2], dtype=np.float32).view(np.int32)[0]
    o.settle_cloth(s, c, 3)
    ee, _ = o.ee_pose(s)
    assert np.isfinite(c).all() and np.abs(c[0, t['anchors']].mean(0) - ee).max() < 0.03                        # the anchors follow the (falling) gripper
    so, se = st[0].copy(), st[0].copy()
    rng = np.random.RandomState(2)
    for k in range(2):
        a = rng.uniform(-1, 1, 5).astype(np.float32)
        o_obs, o_rew, o_done, o_info = o.step(so, a)
        e_obs, e_rew, e_done, e_info, _ = e.step(se, a)
        assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4 and o_done == e_done
        assert np.abs(b.view(so.reshape(1, -1))['q'] - b.view(se.reshape(1, -1))['q']).max() < 5e-5


# ---- the reset of a robot on wheels on the device (csrc/agx_reset.h, AGX_X_FLAGS bit 3) against its numpy restatement ---------------------
@pytest.mark.parametrize('model', ['feeding_stretch', 'scratch_itch_stretch', pytest.param('dressing_stretch', marks=full)])
@pytest.mark.parametrize('seed', [77, (1 << 33) + 5])
def test_device_reset_generator_matches_its_numpy_restatement(model, seed):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.model import compiler as L
    from emu_lib import Emu
    from oracle_lib import Oracle
    from test_reset_generator import assert_same_record
    b = ModelBlob.load(model)
    assert b.has_reset_generator and b.i[b.h['OFF_RESET'] + L.X_['FLAGS']] & 8
    st, info = ro.with_collision_check(b.words).sample(seed)
    se, ie = Emu(b).sample(seed)
    assert_same_record(b, st, se, '%s seed %d' % (model, seed))
    assert info['ik_ok'] and bool(ie[0])
    v = b.view(st.reshape(1, -1))
    d = v['base'][0, :3] - np.array(b.meta['mobile_base'])
    assert np.all(np.abs(d[:2]) <= 0.1 + 1e-6) and abs(d[2]) < 1e-6                                               # env.py:285-286
    yaw = 2 * np.arctan2(v['base'][0, 5], v['base'][0, 6])
    dyaw = (yaw - b.meta['mobile_rpy'][2] + np.pi) % (2 * np.pi) - np.pi
    assert abs(dyaw) <= (np.deg2rad(30) if b.meta.get('mobile_yaw', True) else 0) + 1e-6                          # env.py:287-291
    q = v['q'][0]
    assert abs(q[8] - b.meta['lift']) <= 0.1 + 1e-6 and np.all(q[:8] == 0) and np.all(q[9:14] == 0)               # stretch.py:58-62
    # the tool (or, dressing, the garment) sits at the end effector the oracle's kinematics find for that record
    ee, _ = Oracle(b).ee_pose(st.copy())
    if b.nfree:
        from assistive_gym_amd.host.kin import RobotKin
        tp, _ = RobotKin(b).tool_pose(v['base'][0, :3].astype(np.float64), v['base'][0, 3:].astype(np.float64), q[:b.nrobot].astype(np.float64))
        assert np.linalg.norm(v['free'][0, b.h['TOOL_BODY'], :3] - tp) < 0.2                                       # (COM frame vs base frame of the tool: centimetres)
    else:
        assert np.allclose(v['task'][0, L.DR['CLOTH_OFF']:L.DR['CLOTH_OFF'] + 3].view(np.float32), ee - np.array(b.meta['cloth_orig_pos']), atol=1e-5)


def test_device_reset_redraws_a_colliding_placement():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from assistive_gym_amd.blob import ModelBlob
    b = ModelBlob.load('scratch_itch_stretch')
    calls = []

    def collides(st):
        calls.append(1)
        return len(calls) == 1
    st, info = ro.ResetOracle(b.words, collides).sample(123)
    plain, _ = ro.ResetOracle(b.words).sample(123)
    assert info['rejected_restarts'] == [0] and len(calls) == 2
    v0, v1 = b.view(plain.reshape(1, -1)), b.view(st.reshape(1, -1))
    assert not np.array_equal(v0['base'], v1['base']) and np.array_equal(v0['human'], v1['human']) and v0['q'][0, 8] != v1['q'][0, 8]
