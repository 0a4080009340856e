"""BedBathingEnv.reset on the device (bed_bathing.py:112-171; BASELINE config 3): the sampler of the rag-doll model (drop record), and the
bed-bathing sampler reading the human's resting pose from that model's settled record -- the kernel source (csrc/agx_reset.h) on the wave
emulator and through the C ABI on the GPU against the numpy float64 restatement (oracle/reset_oracle.py)."""
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


@pytest.fixture(scope='module')
def sb():
    from emu_lib import Emu
    b = ModelBlob.load('bed_settle')
    return b, Emu(b)


@pytest.mark.parametrize('seed,imp', [(31, -1), ((1 << 34) + 7, 1)])
def test_ragdoll_drop_record_matches_restatement(sb, seed, imp):
    b, emu = sb
    assert b.has_reset_generator and b.i[b.h['OFF_RESET'] + L.X_['FLAGS']] == 32
    st, info = ro.ResetOracle(b.words).ragdoll_drop(seed, impairment_mode=imp)
    se, ie = emu.sample(seed, impairment_mode=imp)
    assert_same_record(b, st, se, 'seed %d' % seed)
    v = b.view(st.reshape(1, -1))
    q = v['q'][0]
    assert np.allclose(q[:6], [-0.15, 0.2, 0.95, 0, 0, -np.pi / 2], atol=1e-6)                 # bed_bathing.py:121
    assert np.all(np.abs(q[6:]) <= 0.1 + 1e-6) and len(np.unique(np.round(q[6:], 6))) > 30      # :126 (clamped where a limit is closer than the draw)
    assert np.array_equal(v['qt'][0], q) and np.all(v['qd'][0] == 0)
    # the same joint angles the host sampler's clamp produces from these draws
    from assistive_gym_amd.model.human import HumanModel
    hm = HumanModel('female' if info['gender'] else 'male', info['limit_scale'])
    hq = np.zeros(hm.n)
    for k, j in enumerate(b.meta['settle_joints']):
        hq[j] = q[6 + k]
    assert np.allclose(hm.clamp(hq), hq, atol=1e-6)


@pytest.mark.parametrize('robot', ['sawyer', pytest.param('pr2', marks=full), pytest.param('stretch', marks=full)])
def test_bed_bathing_sampler_reads_the_settled_record(sb, robot):
    """settle the drop record on the ORACLE (a few steps are enough for a pose that differs from the drop), then the bed-bathing sampler on
    the emulator against the restatement, both from that record"""
    from emu_lib import Emu
    from oracle_lib import Oracle
    sblob, semu = sb
    b = ModelBlob.load('bed_bathing_' + robot)
    assert b.has_reset_generator and b.i[b.h['OFF_RESET'] + L.X_['FLAGS']] & 16
    seed = 4242
    drop, dinfo = ro.ResetOracle(sblob.words).ragdoll_drop(seed)
    rec = drop.copy()
    Oracle(sblob).settle(rec, 12)
    st, info = ro.with_collision_check(b.words).sample(seed, settled=rec)
    se, ie = Emu(b).sample(seed, settled=rec)
    assert_same_record(b, st, se, robot)
    assert info['gender'] == dinfo['gender'] and info['limit_scale'] == dinfo['limit_scale']      # the two samplers draw the same human
    v = b.view(st.reshape(1, -1))
    # the human's collision bodies are where the rag doll lies: the host path's kinematics from the same record agree
    from assistive_gym_amd.host.reset_bed import settled_pose
    from assistive_gym_amd.model.human import HumanModel
    hm = HumanModel('female' if info['gender'] else 'male', info['limit_scale'])
    bp, bq, hq = settled_pose(sblob, rec.reshape(1, -1), hm)
    hpos, hquat = hm.fk(bp, bq, hq)
    for k, link in enumerate(b.meta['human_bodies']):
        want = bp if link < 0 else hpos[link]
        assert np.allclose(v['human'][0, k, :3], want, atol=2e-5), (k, link)
    assert np.allclose(v['q'][0, b.nrobot:], [hq[j] for j in b.meta['human_dynamic_joints']], atol=1e-6)
    # all targets alive, counted
    nt = b.task_i_n('NT', 4)[2 * info['gender']] + b.task_i_n('NT', 4)[2 * info['gender'] + 1]
    assert v['total_food'][0] == nt
    alive = v['task'][0, L.BB['ALIVE']:L.BB['ALIVE'] + L.BB['ALIVE_WORDS']].view(np.uint32)
    assert sum(bin(int(w)).count('1') for w in alive) == nt
    if robot != 'stretch':
        assert info['ik_ok'] and info['toc']['goals_reached'] >= 1
        from oracle_lib import Oracle as O2
        ee, _ = O2(b).ee_pose(st.copy())
        assert np.linalg.norm(ee - info['target_ee']) < 0.03
    assert v['frozen'][0] == (0 if info['impairment'] == 3 else (((1 << b.nhdof) - 1) << b.nrobot))            # bed_bathing.py:134 + human.py:108


def test_sampler_refuses_without_a_settled_record():
    b = ModelBlob.load('bed_bathing_sawyer')
    with pytest.raises(AssertionError):
        ro.ResetOracle(b.words).sample(5)


@pytest.mark.gpu
def test_gpu_bed_bathing_reset_pipeline():
    """through the C ABI: (1) the rag-doll model's sampler against the restatement; (2) agx_sample_reset of BedBathingSawyer with the rag-doll
    model attached = drop, 100-step settle, sampling from the settled records: every stage against the restatement run on the DEVICE's settled
    records (the settle itself is the stepper's, tested elsewhere); (3) without the attachment the sampler refuses; (4) BASELINE config 3 with
    reset='device': a new human, placement and target set for every environment at the episode boundary"""
    import torch
    from assistive_gym_amd import libagx
    from assistive_gym_amd.libagx import AgxError, Stepper
    from assistive_gym_amd.vec_env import BedBathingSawyerVecEnv, RAGDOLL_SETTLE_STEPS, build_reset_pool
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    sblob, blob = ModelBlob.load('bed_settle'), ModelBlob.load('bed_bathing_sawyer')
    n = 6
    rag = Stepper(sblob, n)
    rag.sample_reset(7001)
    rag.synchronize()
    drops = rag.get_state()
    so = ro.ResetOracle(sblob.words)
    for i in range(n):
        assert_same_record(sblob, so.ragdoll_drop(7001 + i)[0], drops[i], 'drop %d' % i)
    st = Stepper(blob, n)
    with pytest.raises(AgxError):
        st.sample_reset(7001)
    st.attach_settle_model(rag, RAGDOLL_SETTLE_STEPS)
    st.sample_reset(7001)
    st.synchronize()
    got, settled = st.get_state(), rag.get_state()
    o = ro.with_collision_check(blob.words)
    for i in range(n):
        v = sblob.view(settled[i:i + 1])
        assert 0.7 < v['q'][0, 2] < 0.93                                                    # came down onto the bed (dropped from z = 0.95 under gravity -1); limbs may still move: the reset zeroes the velocities (bed_bathing.py:136-137)
        want, info = o.sample(7001 + i, settled=settled[i])
        assert_same_record(blob, want, got[i], 'env %d' % i)
    st.close(); rag.close()
    # the pool the product builds: finite, humans differ, robots placed beside the bed
    pool = build_reset_pool(blob, 16, 8001)
    v = blob.view(pool)
    assert np.isfinite(pool[:, :blob.h['S_ENV']]).all() and len(np.unique(np.round(v['human'][:, 0, 0], 5))) > 8
    env = BedBathingSawyerVecEnv(32, reset='device', seed=23)
    obs = env.reset()
    v0 = env.blob.view(env.stepper.get_state().copy())
    assert torch.isfinite(obs).all()
    for k in range(200):
        obs, rew, done, info = env.step(torch.zeros((32, env.act_dim), device='cuda'))
        assert bool(done.all()) == (k == 199)
    v1 = env.blob.view(env.stepper.get_state())
    assert torch.isfinite(obs).all() and (v1['iteration'] == 0).all()
    assert (np.abs(v1['human'][:, 0, :3] - v0['human'][:, 0, :3]).max(axis=1) > 1e-6).all()     # a NEW human for every environment
    env.close()
