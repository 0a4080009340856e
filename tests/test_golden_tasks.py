"""Committed oracle trajectories of one model per task (tests/golden/*_oracle_traj.npz, written by tests/diag/make_golden_tasks.py:
a host-sampled post-reset state, 12 random actions, the oracle's observations / rewards / final state).  Regression fixtures that freeze
oracle and device together -- NOT reference data.  CPU: the oracle reproduces them exactly, the kernel sources on the wave emulator
replay the first steps free-running; GPU: the whole trajectory free-running through the C ABI."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(os.path.basename(p)[:-len('_oracle_traj.npz')] for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', '*_oracle_traj.npz'))
                if not os.path.basename(p).startswith('feeding_jaco'))


def _load(name):
    from assistive_gym_amd.blob import ModelBlob
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '_oracle_traj.npz'))
    coop = bool(g['coop'])
    b = ModelBlob.load(name[:-5] if coop else name)
    return (b.coop() if coop else b), g


def _force_columns(b):
    """observation entries that are contact forces (1e-3 relative): the last one or two of the robot part, the last two or three of the human part"""
    from assistive_gym_amd.model import compiler as L
    nf = 2 if b.task_kind == L.TASK_ARM_MANIPULATION else (2 if b.task_kind == L.TASK_FEEDING else 1)
    r = b.obs_dim_robot
    cols = list(range(r - nf, r))
    if b.is_coop:
        cols += list(range(b.obs_dim - (3 if b.task_kind == L.TASK_ARM_MANIPULATION else 2), b.obs_dim))
    return cols


def _check(b, obs, rew, g, k, tol):
    f = _force_columns(b)
    dev = np.abs(obs - g['obs'][k])
    assert np.all(dev[f] <= 1e-3 * np.maximum(1.0, np.abs(g['obs'][k][f])) + tol), (k, dev[f])
    dev[f] = 0
    assert dev.max() < tol and abs(float(rew) - float(g['reward'][k])) < tol * max(1.0, abs(float(g['reward'][k]))) + 1e-2 * 1e-3 * np.abs(g['obs'][k][f]).sum(), (k, dev.max(), rew, g['reward'][k])


@pytest.mark.parametrize('name', GOLDEN)
def test_oracle_reproduces_the_fixture(name):
    from oracle_lib import Oracle
    b, g = _load(name)
    o = Oracle(b)
    s = g['state0'].copy()
    for k, a in enumerate(g['actions']):
        obs, rew, done, info = o.step(s, a)
        assert np.array_equal(obs, g['obs'][k]) and rew == g['reward'][k], k
    assert np.array_equal(s.view(np.uint32), g['state_end'].view(np.uint32))


@pytest.mark.parametrize('name', GOLDEN)
def test_emulator_replays_the_first_steps(name):
    from emu_lib import Emu
    b, g = _load(name)
    e = Emu(b)
    s = g['state0'].copy()
    for k in range(2):
        obs, rew, done, info, _ = e.step(s, g['actions'][k])
        _check(b, obs, rew, g, k, 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('name', GOLDEN)
def test_gpu_replays_the_trajectory(name):
    from assistive_gym_amd import libagx
    from assistive_gym_amd.libagx import Stepper
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    b, g = _load(name)
    st = Stepper(b, 1)
    st.set_state(g['state0'][None])
    for k in range(len(g['actions'])):
        obs, rew, done, info = st.step_host(g['actions'][k][None])
        _check(b, obs[0], rew[0], g, k, 5e-4)
    v, w = b.view(st.get_state()), b.view(g['state_end'][None].copy())
    assert np.abs(v['q'] - w['q']).max() < 5e-4 and v['iteration'][0] == w['iteration'][0]
    st.close()
