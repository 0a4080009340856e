"""ArmManipulationJaco-v1 / ArmManipulationPanda-v1 (arm_manipulation_envs.py:27-37: single-arm robots on a nightstand beside the bed)
without a GPU: blobs against the reference's robot tables, the host reset with the device code's collision pass (wave emulator) and
the arm_manipulation kernel variant on the emulator against the oracle.  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

from conftest import full
from test_scratch_itch_robots import emu_checker, flags_from_oracle


@pytest.fixture(scope='module', params=['jaco', pytest.param('panda', marks=full)])
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


def test_model_tables(rb):
    from assistive_gym_amd.model import compiler as L
    name, b, o, e, fall = rb
    T = L.robot_table('arm_manipulation', name)
    assert b.task_kind == L.TASK_ARM_MANIPULATION and (b.act_dim, b.obs_dim) == (14, 45) and b.task_i('DUP_ACT') == 7
    arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
    assert [b.robot_i(d, 'PB_INDEX') for d in arm_dofs] == T['arm'] and [b.robot_i(d, 'ACT') for d in arm_dofs] == list(range(7, 14))
    grip_dofs = [d for d in range(b.nrobot) if b.robot_i(d, 'PB_INDEX') in T['grip']]
    assert np.allclose([b.robot_f(d, 'QT0') for d in grip_dofs], T['gripper_target'])
    assert np.isclose(b.robot_f(arm_dofs[0], 'MAXF'), 20.0) and np.isclose(b.param('HUMAN_GRAVITY_Z'), -9.81)
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (24, 87)


def test_reset_with_collision_rejection_and_emulator_parity(rb):
    name, b, o, e, fall = rb
    n = 6
    raw, _ = _states(b, n, 1001, arm_settler=fall)
    want = np.array([flags_from_oracle(b, o, s) for s in raw])
    assert np.array_equal(emu_checker(e)(raw), want)
    st, infos = _states(b, n, 1001, arm_settler=fall, checker=emu_checker(e))
    after = np.array([flags_from_oracle(b, o, s) for s in st])
    assert np.array_equal(after, [i['collision_flags'] for i in infos]) and (after != 0).sum() <= max(1, (want != 0).sum() // 2), (want, after)
    assert np.isfinite(st).all()
    v0, v1 = b.view(raw), b.view(st)
    assert np.array_equal(v0['q'][:, b.nrobot:], v1['q'][:, b.nrobot:])          # the fallen arm is not touched by the re-draws
    clean = np.flatnonzero(after == 0)[:2]
    for i in clean:
        so, se = st[i].copy(), st[i].copy()
        for k in range(3):
            a = np.random.RandomState(10 * int(i) + k).uniform(-1, 1, 14).astype(np.float32)
            oo, orr, od, oi = o.step(so, a)
            eo, er, ed, ei, _ = e.step(se, a)
            assert oi[6] == ei[6] and oi[7] == ei[7] and oi[4] == ei[4]
            assert np.abs(oo - eo).max() < 2e-5 * max(1.0, np.abs(oo).max()) and abs(orr - er) < 2e-5 * max(1.0, abs(orr))
