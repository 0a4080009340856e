"""Device-side reset generator, base pose search: ScratchItchEnv.reset for the free-standing PR2 / Baxter (scratch_itch.py:93-132 with
Robot.position_robot_toc, robot.py:123-228: 50 candidate base poses, one per lane; IK for the start pose and the three position goals on
the human's arm per candidate; goals reached, then summed JLWKI decide) -- the kernel source (csrc/agx_reset.h) on the wave emulator and
through the C ABI on the GPU against its numpy float64 restatement (oracle/reset_oracle.py: own Philox, numpy.linalg solve / det)."""
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


@pytest.fixture(scope='module', params=[pytest.param('pr2', marks=full), pytest.param('baxter', marks=full), 'sawyer'])
def rb(request):
    from emu_lib import Emu
    b = ModelBlob.load('scratch_itch_' + request.param)
    assert b.has_reset_generator and b.meta['mount'] == 'toc'
    return request.param, b, Emu(b)


def _x(blob, key, as_int=False):
    o = blob.h['OFF_RESET'] + L.X_[key]
    return int(blob.i[o]) if as_int else float(blob.f[o])


@pytest.mark.parametrize('seed', [77, (1 << 33) + 5])
def test_emulated_kernel_matches_oracle(rb, seed):
    name, blob, emu = rb
    o = ro.with_collision_check(blob.words)
    st, info = o.sample(seed)
    se, ie = emu.sample(seed)
    assert_same_record(blob, st, se, '%s seed %d' % (name, seed))
    assert info['ik_ok'] and bool(ie[0]) and int(ie[1]) == info['ik_restarts'] and int(ie[2]) == info['toc']['goals_reached'] >= 1
    v = blob.view(st.reshape(1, -1))
    assert ro.ResetOracle(blob.words).chain() == ([0, 2, 3, 4, 5, 6, 7] if name == 'sawyer' else list(range(7)))      # the Sawyer's chain skips its head pan
    # the chosen base lies in the sampled box on the human's right, turned by at most 30 degrees (robot.py:142-146, env.py:298)
    base0 = blob.f[blob.h['OFF_RESET'] + L.X_['BASE_POS']:blob.h['OFF_RESET'] + L.X_['BASE_POS'] + 3].astype(np.float64)
    d = v['base'][0, :3] - base0
    assert -0.5 <= d[0] <= 1e-6 and abs(d[1]) <= 0.5 + 1e-6 and abs(d[2]) < 1e-6
    assert abs(2 * np.arctan2(v['base'][0, 5], v['base'][0, 6])) <= np.deg2rad(30) + 1e-6 and v['base'][0, 3] == 0 and v['base'][0, 4] == 0
    # the start pose is reached: the end effector sits within the threshold of the drawn start position
    from oracle_lib import Oracle
    ee, _ = Oracle(blob).ee_pose(st.copy())
    assert np.linalg.norm(ee - info['target_ee']) < 0.03


def test_nobody_reaches_the_start_pose(rb):
    """a threshold no candidate can meet: all four rounds are spent, the record is still a valid world (candidate 0 of the last round)"""
    from emu_lib import Emu
    name, blob, _ = rb
    w = blob.words.copy(); w.view(np.float32)[blob.h['OFF_RESET'] + L.X_['TOC_THRESH']] = 1e-12
    w.view(np.int32)[blob.h['OFF_RESET'] + L.X_['TOC_ATTEMPTS']] = 6                 # (six candidates: keeps the numpy side quick)
    b = ModelBlob(w, blob.meta)
    st, info = ro.ResetOracle(b.words).sample(5)
    se, ie = Emu(b).sample(5)
    assert not info['ik_ok'] and info['ik_restarts'] == 4 and not bool(ie[0]) and int(ie[1]) == 4
    assert_same_record(b, st, se)
    assert np.isfinite(st).all()


def test_a_colliding_placement_is_drawn_again(rb):
    """with the collision verdict forced to "collides" once, the second placement uses its own streams: another base pose"""
    name, blob, emu = rb
    calls = []

    def collides(st):
        calls.append(1)
        return len(calls) == 1
    o = ro.ResetOracle(blob.words, collides)
    st, info = o.sample(123)
    plain, _ = ro.ResetOracle(blob.words).sample(123)
    assert info['rejected_restarts'] == [0] and len(calls) == 2
    v0, v1 = blob.view(plain.reshape(1, -1)), blob.view(st.reshape(1, -1))
    assert not np.array_equal(v0['base'], v1['base']) and np.array_equal(v0['human'], v1['human'])


@pytest.mark.gpu
def test_gpu_base_pose_search_matches_oracle():
    import torch
    from assistive_gym_amd import libagx
    from assistive_gym_amd.libagx import Stepper
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    blob = ModelBlob.load('scratch_itch_pr2')
    o = ro.with_collision_check(blob.words)
    n = 6
    st = Stepper(blob, n)
    info = torch.zeros((n, 4), device='cuda')
    st.sample_reset(9001, ik_info=info)
    st.synchronize()
    got, gi = st.get_state(), info.cpu().numpy()
    for i in range(n):
        so, io = o.sample(9001 + i)
        assert_same_record(blob, so, got[i], 'env %d' % i)
        assert bool(gi[i, 0]) == io['ik_ok'] and int(gi[i, 1]) == io['ik_restarts']
    st.close()


@pytest.mark.gpu
def test_gpu_config4_resets_on_the_device():
    """ScratchItchPR2Human-v1 (BASELINE config 4) with reset='device': every episode of every environment starts from a newly sampled human,
    target and robot placement (VERDICT r2 item 5 for this config); the batch keeps stepping across the boundary"""
    import torch
    from assistive_gym_amd import libagx
    from assistive_gym_amd.vec_env import ScratchItchPR2HumanVecEnv
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    n = 64
    env = ScratchItchPR2HumanVecEnv(n, reset='device', seed=21)
    obs = env.reset()
    first = env.stepper.get_state().copy()
    v0 = env.blob.view(first)
    assert torch.isfinite(obs).all() and len(np.unique(np.round(v0['base'][:, 0], 5))) > n // 2          # placements differ between environments
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
        assert bool(done.all()) == (k == 199)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    v1 = env.blob.view(env.stepper.get_state())
    assert (np.abs(v1['base'][:, :2] - v0['base'][:, :2]).max(axis=1) > 1e-4).all()                      # a NEW placement for every environment
    assert (v1['iteration'] == 0).all()
    env.close()


# ---- DressingEnv.reset on the device (dressing.py:112-198): oriented goals above the human's left arm, the garment words ----------------
@pytest.fixture(scope='module', params=['baxter', 'jaco'])
def dr(request):
    from emu_lib import Emu
    b = ModelBlob.load('dressing_' + request.param)
    assert b.has_reset_generator
    return request.param, b, Emu(b)


def test_dressing_emulated_kernel_matches_oracle(dr):
    name, blob, emu = dr
    st, info = ro.with_collision_check(blob.words).sample(31)
    se, ie = emu.sample(31)
    assert_same_record(blob, st, se, name)
    assert info['ik_ok'] and bool(ie[0])
    v = blob.view(st.reshape(1, -1))
    tw = v['task'][0].view(np.float32)
    assert tw[L.DR['CLOTH_GRAVITY']] == np.float32(-9.81 / 2)                                              # dressing.py:178
    from oracle_lib import Oracle
    ee, _ = Oracle(blob).ee_pose(st.copy())
    assert np.abs(tw[L.DR['CLOTH_OFF']:L.DR['CLOTH_OFF'] + 3] - (ee - np.array(blob.meta['cloth_orig_pos']))).max() < 1e-5     # dressing.py:146-149
    if name == 'baxter':      # the robot stands on the human's LEFT, turned by pi +- 30 degrees (env.py:298, robot.py:143)
        base0 = blob.f[blob.h['OFF_RESET'] + L.X_['BASE_POS']:blob.h['OFF_RESET'] + L.X_['BASE_POS'] + 3]
        d = v['base'][0, :3] - base0
        assert -1e-6 <= d[0] <= 0.5 and abs(d[1]) <= 0.5 + 1e-6
        yaw = 2 * np.arctan2(v['base'][0, 5], v['base'][0, 6])
        assert abs(abs(yaw) - np.pi) <= np.deg2rad(30) + 1e-6
        assert info['toc']['goals_reached'] >= 1


@pytest.mark.gpu
def test_gpu_dressing_resets_on_the_device():
    """DressingBaxter-v1 (BASELINE config 5) with reset='device': agx_reset samples the scene, loads the garment at the end effector, settles it
    for 50 steps under half gravity and switches to full gravity; the episode boundary does the same for every environment"""
    import torch
    from assistive_gym_amd import libagx
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import DressingBaxterVecEnv
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    blob = ModelBlob.load('dressing_baxter')
    oc = blob.h['OFF_CLOTH']
    nn = int(blob.i[oc + L.CL['NN']])
    x0 = blob.f[oc + int(blob.i[oc + L.CL['OFF_X0']]):oc + int(blob.i[oc + L.CL['OFF_X0']]) + 3 * nn].reshape(nn, 3)
    # sampling alone: the record against the numpy restatement, the garment = X0 + offset at rest
    o = ro.with_collision_check(blob.words)
    st = Stepper(blob, 3)
    st.sample_reset(500)
    st.synchronize()
    got, cloth = st.get_state(), st.get_cloth()
    for i in range(3):
        so, io = o.sample(500 + i)
        assert_same_record(blob, so, got[i], 'env %d' % i)
        off = blob.view(got[i:i + 1])['task'][0].view(np.float32)[L.DR['CLOTH_OFF']:L.DR['CLOTH_OFF'] + 3]
        assert np.abs(cloth[i, 0] - (x0 + off)).max() < 1e-6 and not cloth[i, 1].any()
    st.close()
    # the whole reset in a VecEnv: settled garments hanging below the end effector, full gravity, fresh states at the boundary
    n = 8
    env = DressingBaxterVecEnv(n, reset='device', seed=77)
    obs = env.reset()
    s0, c0 = env.stepper.get_state(), env.stepper.get_cloth()
    v = env.blob.view(s0)
    assert torch.isfinite(obs).all() and np.isfinite(c0).all()
    assert (v['task'][:, L.DR['CLOTH_GRAVITY']].view(np.float32) == np.float32(-9.81)).all() and (v['iteration'] == 0).all()
    assert np.percentile(np.linalg.norm(c0[:, 1], axis=2), 90) < 1.5                                      # settled
    assert (c0[:, 0, :, 2].min(axis=1) < c0[:, 0, :, 2].max(axis=1) - 0.5).all()                          # hanging: more than half a metre tall
    a = torch.zeros(n, 7, device='cuda')
    for k in range(200):
        obs, rew, done, info = env.step(a)
        assert bool(done.all()) == (k == 199)
    s1, c1 = env.stepper.get_state(), env.stepper.get_cloth()
    v1 = env.blob.view(s1)
    assert torch.isfinite(obs).all() and np.isfinite(c1).all() and (v1['iteration'] == 0).all()
    assert (np.abs(v1['base'][:, :2] - v['base'][:, :2]).max(axis=1) > 1e-4).all()                        # a NEW placement for every environment
    assert (v1['task'][:, L.DR['CLOTH_GRAVITY']].view(np.float32) == np.float32(-9.81)).all()
    env.close()


# ---- FeedingEnv.reset for the free-standing robots (feeding.py:139-142: the mouth is the one goal besides the start pose); the Sawyer's pedestal guard
@pytest.mark.parametrize('robot', ['sawyer', 'baxter'])
def test_feeding_base_pose_search_matches_oracle(robot):
    from emu_lib import Emu
    blob = ModelBlob.load('feeding_' + robot)
    assert blob.has_reset_generator and _x(blob, 'TOC_NGOALS', True) == 1 and _x(blob, 'TOC_GOAL_KIND', True) == 1
    st, info = ro.with_collision_check(blob.words).sample(1005)
    se, ie = Emu(blob).sample(1005)
    assert_same_record(blob, st, se, robot)
    assert info['ik_ok'] and info['toc']['goals_reached'] == 2                       # start pose and the mouth
    v = blob.view(st.reshape(1, -1))
    assert np.linalg.norm(v['target'][0]) > 0.5 and v['free'][0, blob.h['FOOD0'], 2] > 0.9          # the mouth target and the food above the spoon exist


def test_every_feeding_scratch_and_dressing_model_resets_on_the_device():
    for task in ('feeding', 'scratch_itch', 'dressing'):
        for robot in ('jaco', 'panda', 'sawyer', 'baxter', 'pr2'):
            b = ModelBlob.load('%s_%s' % (task, robot))
            assert b.has_reset_generator, (task, robot)
            mounted = b.meta['mount'] == 'wheelchair'
            assert (_x(b, 'TOC_ATTEMPTS', True) == 0) == mounted and _x(b, 'PED_N', True) == (2 if robot == 'sawyer' else 0)
    for robot in ('jaco', 'panda', 'sawyer', 'baxter', 'pr2', 'stretch'):             # bed bathing: with the rag-doll model attached (tests/test_reset_bed_device.py)
        b = ModelBlob.load('bed_bathing_' + robot)
        assert b.has_reset_generator and _x(b, 'FLAGS', True) & 16
    for robot in ('sawyer', 'jaco', 'panda', 'pr2', 'baxter'):                           # two settles (rag doll, then the arm's fall): bits 4 and 7, tests/test_reset_arm_device.py
        b = ModelBlob.load('arm_manipulation_' + robot)
        assert b.has_reset_generator and _x(b, 'FLAGS', True) == 1 | 16 | 128 | (512 if robot in ('pr2', 'baxter') else 0) and _x(b, 'TOC_NGOALS', True) == 4


def test_pedestal_guard_rejects_start_poses_inside_the_boxes():
    """the Sawyer's guard as a candidate filter: with boxes that contain the whole workspace nobody is accepted (and the device code agrees);
    with the real boxes some seeds choose another candidate than without the guard"""
    from emu_lib import Emu
    blob = ModelBlob.load('scratch_itch_sawyer')
    o0 = blob.h['OFF_RESET']
    w = blob.words.copy(); f = w.view(np.float32)
    f[o0 + L.X_['PED_BOX']:o0 + L.X_['PED_BOX'] + 6] = [-10, -10, -10, 10, 10, 10]
    w.view(np.int32)[o0 + L.X_['TOC_ATTEMPTS']] = 5
    big = ModelBlob(w, blob.meta)
    st, info = ro.ResetOracle(big.words).sample(3)
    se, ie = Emu(big).sample(3)
    assert not info['ik_ok'] and info['ik_restarts'] == 4 and not bool(ie[0])
    assert_same_record(big, st, se)
    w2 = blob.words.copy(); w2.view(np.int32)[o0 + L.X_['PED_N']] = 0
    w2.view(np.int32)[o0 + L.X_['TOC_ATTEMPTS']] = 12; w3 = blob.words.copy(); w3.view(np.int32)[o0 + L.X_['TOC_ATTEMPTS']] = 12
    off, on = ro.ResetOracle(ModelBlob(w2, blob.meta).words), ro.ResetOracle(ModelBlob(w3, blob.meta).words)
    differ = 0
    for seed in range(40, 52):                       # (stops at the first seed that shows the difference: the numpy base pose search is slow)
        a, b = off.sample(seed)[0], on.sample(seed)[0]
        differ += int(not np.array_equal(a, b))
        if differ:
            break
    assert differ >= 1


@pytest.mark.gpu
@pytest.mark.parametrize('cls,model', [('FeedingSawyerVecEnv', 'feeding_sawyer'), ('ScratchItchSawyerVecEnv', 'scratch_itch_sawyer'), ('FeedingPR2VecEnv', 'feeding_pr2')])
def test_gpu_free_standing_robots_reset_on_the_device(cls, model):
    """the free-standing robots of the feeding / scratch-itch scenes with reset='device' (the Sawyer's chain skips its head pan and its
    candidates pass the pedestal guard): records against the numpy restatement, then episodes across a boundary"""
    import torch
    from assistive_gym_amd import libagx, vec_env
    from assistive_gym_amd.libagx import Stepper
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    blob = ModelBlob.load(model)
    o = ro.with_collision_check(blob.words)
    st = Stepper(blob, 3)
    st.sample_reset(700)
    st.synchronize()
    got = st.get_state()
    for i in range(3):
        so, io = o.sample(700 + i)
        assert_same_record(blob, so, got[i], '%s env %d' % (model, i))
    st.close()
    n = 32
    env = getattr(vec_env, cls)(n, reset='device', seed=9)
    obs = env.reset()
    b0 = env.blob.view(env.stepper.get_state())['base'].copy()
    assert torch.isfinite(obs).all()
    g = torch.Generator(device='cuda'); g.manual_seed(2)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
    assert bool(done.all()) and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    b1 = env.blob.view(env.stepper.get_state())['base']
    assert (np.abs(b1[:, :2] - b0[:, :2]).max(axis=1) > 1e-4).all()
    env.close()
