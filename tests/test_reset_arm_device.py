"""ArmManipulationEnv.reset on the device (arm_manipulation.py:110-180; single-arm robots): three models in a row -- the rag doll's settle
(bed_settle, dropped from [-0.25, 0.2, 0.95]), the arm's fall (the task's blob at gravity -1 with its sampler writing the record the posed
arm falls from: ModelBlob.fall_model()), then the task's own sampler reading both records (bodies, arm angles and velocities from the fall;
the four goals of the base pose search from the tree with the fallen arm).  The kernel source (csrc/agx_reset.h) on the wave emulator
against the numpy float64 restatement (oracle/reset_oracle.py), and the restatement against the host sampler's kinematics."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import reset_oracle as ro                      # noqa: E402  (test infrastructure)
from assistive_gym_amd.blob import ModelBlob   # noqa: E402
from assistive_gym_amd.model import compiler as L   # noqa: E402
from conftest import full                      # noqa: E402
from test_reset_generator import assert_same_record   # noqa: E402

DROP = [-0.25, 0.2, 0.95]


def chain(robot, seed, rag_steps=12, fall_steps=10):
    """rag doll and fall on the ORACLE (a few steps each: enough for poses that differ from the samplers' own), records for the samplers"""
    from oracle_lib import Oracle
    sblob = ModelBlob.load('bed_settle').with_drop_base(DROP)
    b = ModelBlob.load('arm_manipulation_' + robot)
    fb = b.fall_model()
    drop, dinfo = ro.ResetOracle(sblob.words).ragdoll_drop(seed, impairment_mode=ro.MODE_NO_TREMOR)
    rag = drop.copy(); Oracle(sblob).settle(rag, rag_steps)
    return sblob, b, fb, rag, dinfo


@pytest.mark.parametrize('robot', ['sawyer', pytest.param('pr2', marks=full), pytest.param('jaco', marks=full), pytest.param('panda', marks=full), pytest.param('baxter', marks=full)])
def test_fall_record_matches_restatement(robot):
    from emu_lib import Emu
    seed = 7301
    sblob, b, fb, rag, dinfo = chain(robot, seed)
    dual = robot in ('pr2', 'baxter')
    assert b.has_reset_generator and b.i[b.h['OFF_RESET'] + L.X_['FLAGS']] == 1 | 16 | 128 | (512 if dual else 0)
    assert fb.i[fb.h['OFF_RESET'] + L.X_['FLAGS']] & 256 and fb.param('HUMAN_GRAVITY_Z') == -1.0 and b.param('HUMAN_GRAVITY_Z') == pytest.approx(-9.81)
    st, info = ro.ResetOracle(fb.words).sample(seed, impairment_mode=ro.MODE_NO_TREMOR, settled=rag)
    se, ie = Emu(fb).sample(seed, impairment_mode=ro.MODE_NO_TREMOR, settled=rag)
    assert_same_record(fb, st, se, robot)
    assert info['gender'] == dinfo['gender'] and info['limit_scale'] == dinfo['limit_scale'] and info['impairment'] != 3
    v = fb.view(st.reshape(1, -1))
    # the host sampler's record from the same resting pose (host/reset_arm.py arm_fall_record): bodies, posed arm, parked robot, the hold
    from assistive_gym_amd.host.reset_arm import ArmManipulationSawyerReset, PARKED
    from assistive_gym_amd.host.reset_bed import settled_pose
    from assistive_gym_amd.model.human import HumanModel
    hm = HumanModel('female' if info['gender'] else 'male', info['limit_scale'])
    bp, bq, hq = settled_pose(sblob, rag.reshape(1, -1), hm)
    rs = ArmManipulationSawyerReset(b)
    pre = dict(hm=hm, hq=hq, base_pos=bp, base_quat=bq, strength=info['strength'], limit_scale=info['limit_scale'], plane_friction=float(v['plane_friction'][0]), gender='female' if info['gender'] else 'male')
    host = b.new_state(1)
    rs.arm_fall_record(host, pre, env_seed=seed)
    hv = b.view(host)
    assert np.allclose(v['human'][0], hv['human'][0], atol=3e-5)
    assert np.allclose(v['q'][0], hv['q'][0], atol=2e-6) and np.allclose(v['qt'][0], hv['qt'][0], atol=2e-6) and np.all(v['qd'][0] == 0)
    nr = b.nrobot
    dyn = b.meta['human_dynamic_joints']
    assert abs(v['q'][0, nr + dyn.index(3)] - min(np.deg2rad(60), hm.upper[3])) < 1e-6 and v['q'][0, nr + dyn.index(6)] == pytest.approx(max(0.0, hm.lower[6]), abs=1e-6)
    assert np.allclose(v['base'][0, :3], PARKED) and np.allclose(v['free'][0, 0, :3], hv['free'][0, 0, :3], atol=2e-5)
    assert abs(abs(float(v['free'][0, 0, 3:7] @ hv['free'][0, 0, 3:7])) - 1.0) < 1e-6                 # the same orientation (q and -q)
    if dual:                                                                                         # the second scooper in the left hand
        assert np.allclose(v['free'][0, 1, :3], hv['free'][0, 1, :3], atol=2e-5) and abs(abs(float(v['free'][0, 1, 3:7] @ hv['free'][0, 1, 3:7])) - 1.0) < 1e-6
    assert v['frozen'][0] == 0 and v['human_kp'][0] == pytest.approx(0.05) and v['human_maxf'][0] == pytest.approx(0.01 * info['strength']) and v['total_food'][0] == 1
    assert np.allclose(v['tremor_target'][0], v['q'][0, nr:])


@pytest.mark.parametrize('robot', ['sawyer', 'pr2', pytest.param('jaco', marks=full), pytest.param('panda', marks=full), pytest.param('baxter', marks=full)])
def test_post_fall_sampler_reads_both_records(robot):
    from emu_lib import Emu
    from oracle_lib import Oracle
    seed = 7302
    sblob, b, fb, rag, dinfo = chain(robot, seed)
    rec, _ = ro.ResetOracle(fb.words).sample(seed, impairment_mode=ro.MODE_NO_TREMOR, settled=rag)
    fell = rec.copy(); Oracle(fb).settle(fell, 10)
    fv = fb.view(fell.reshape(1, -1))
    nr = b.nrobot
    assert np.abs(fv['q'][0, nr:] - fb.view(rec.reshape(1, -1))['q'][0, nr:]).max() > 1e-3 and np.abs(fv['qd'][0, nr:]).max() > 1e-3       # the arm is falling
    st, info = ro.with_collision_check(b.words).sample(seed, impairment_mode=ro.MODE_NO_TREMOR, settled=rag, fell=fell)
    se, ie = Emu(b).sample(seed, impairment_mode=ro.MODE_NO_TREMOR, settled=rag, fell=fell)
    assert_same_record(b, st, se, robot)
    v = b.view(st.reshape(1, -1))
    assert np.array_equal(v['human'][0], fv['human'][0])
    for key in ('q', 'qd', 'qt'):
        assert np.array_equal(v[key][0, nr:], fv[key][0, nr:]), key
    assert np.array_equal(v['tremor_target'][0], fv['tremor_target'][0]) and np.all(v['qd'][0, :nr] == 0)
    assert v['frozen'][0] == 0 and v['human_kp'][0] == pytest.approx(0.05) and v['total_food'][0] == 1 and v['iteration'][0] == 0
    assert info['ik_ok'] and info['toc']['goals_reached'] >= 1
    assert np.abs(v['base'][0, :2] - fv['base'][0, :2]).max() > 5.0                          # the robot has left its parking spot
    # the four goals of the base pose search are the host sampler's: FK of the tree with the fallen arm (host/reset_arm.py post_fall)
    from assistive_gym_amd.host.reset_bed import settled_pose
    from assistive_gym_amd.model.human import HumanModel
    hm = HumanModel('female' if info['gender'] else 'male', info['limit_scale'])
    bp, bq, hq = settled_pose(sblob, rag.reshape(1, -1), hm)
    for k, j in enumerate(b.meta['human_dynamic_joints']):
        hq[j] = fv['q'][0, nr + k]
    hpos, _ = hm.fk(bp, bq, hq)
    R = ro.ResetOracle(b.words)
    R.settled = np.asarray(rag, dtype=np.float32)[:6 + 41].astype(np.float64); R.fell = fell.astype(np.float64)
    for link in (9, 27, 7, 24):
        assert np.allclose(R.link_pose(info['gender'], link, info['limit_scale'], [0, 0, 0])[0], hpos[link], atol=3e-5), link
    x0 = b.h['OFF_RESET']
    assert list(b.i[x0 + L.X_['TOC_GOAL_LINKS']:x0 + L.X_['TOC_GOAL_LINKS'] + 3]) + [int(b.i[x0 + L.X_['TOC_GOAL_LINK3']])] == [9, 27, 7, 24]
    # the tool sits in the gripper of the placed arm; the end effector reached its start target
    assert np.linalg.norm(v['free'][0, 0, :3] - v['base'][0, :3]) < 1.6
    if robot in ('pr2', 'baxter'):                      # both start poses reached: the two scoopers near their targets (arm_manipulation.py:158-159)
        assert np.linalg.norm(v['free'][0, 0, :3] - np.array([-1, -0.3, 0.8])) < 0.35 and np.linalg.norm(v['free'][0, 1, :3] - np.array([-1, 0.7, 0.8])) < 0.35
