"""FeedingSawyer-v1 and FeedingBaxter-v1 (feeding_envs.py:21-27: free-standing robots, robot_arm = 'right') without a GPU: blobs against
the reference's robot tables, the host reset (base pose search + the device code's collision pass, here on the wave emulator) and the
feeding_l kernel variant (320 colliders) on the emulator against the oracle.  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X
from conftest import full
from test_scratch_itch_robots import emu_checker, flags_from_oracle


@pytest.fixture(scope='module', params=['sawyer', pytest.param('baxter', marks=full), pytest.param('pr2', marks=full)])
def rb(request):
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    from oracle_lib import Oracle
    b = ModelBlob.load('feeding_' + request.param)
    return request.param, b, Oracle(b), Emu(b)


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset import make_states
    return make_states(blob, n, seed=seed, **kw)


def test_model_tables(rb):
    from assistive_gym_amd.model import compiler as L
    name, b, o, e = rb
    T = L.FEEDING_ROBOTS[name]
    assert b.task_kind == L.TASK_FEEDING and (b.act_dim, b.obs_dim, b.nhdof, b.nfood) == (7, 25, 4, 8)
    arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
    assert [b.robot_i(d, 'PB_INDEX') for d in arm_dofs] == T['arm']
    assert np.isclose(b.robot_f(arm_dofs[0], 'KP'), 0.025) and np.isclose(b.robot_f(arm_dofs[0], 'MAXF'), 1.0)     # feeding.py:122, robot.py:36
    assert b.meta['mount'] == 'toc' and b.h['NCOLL'] > 256                                                          # needs the feeding_l variant
    assert b.i[b.h['OFF_RESET'] + L.X_['NARM']] == 7 and b.i[b.h['OFF_RESET'] + L.X_['TOC_ATTEMPTS']] == 50        # the device-side reset generator searches the base pose (robot.py:123)
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (11, 25 + 23)


def test_reset_with_collision_rejection(rb):
    name, b, o, e = rb
    n = 8
    raw, infos0 = _states(b, n, 2001)
    want_q = X.quat_from_rpy(b.meta['ee_rpy'])
    for i in range(n):
        p, q = o.ee_pose(raw[i])
        assert np.linalg.norm(p - infos0[i]['target_ee_pos']) < 0.031 and min(np.linalg.norm(q - want_q), np.linalg.norm(q + want_q)) < 0.031
        v = b.view(raw[i:i + 1])
        spoon, food = v['free'][0, 0, :3], v['free'][0, 2:, :3]
        assert np.all(np.linalg.norm(food - spoon, axis=1) < 0.2)
        # the base stands within the search window around [-0.85, -0.4, 0] + toc_base (robot.py:141-142)
        d = v['base'][0, :3] - (np.array([-0.85, -0.4, 0]) + b.meta['toc_base'])
        assert -0.5 - 1e-6 <= d[0] <= 1e-6 and abs(d[1]) <= 0.5 + 1e-6 and abs(d[2]) < 1e-6
    got = emu_checker(e)(raw)
    want = np.array([flags_from_oracle(b, o, s) for s in raw])
    assert np.array_equal(got, want)
    st, infos = _states(b, n, 2001, checker=emu_checker(e))
    after = np.array([flags_from_oracle(b, o, s) for s in st])
    assert np.array_equal(after, [i['collision_flags'] for i in infos]) and (after != 0).sum() <= max(1, (want != 0).sum() // 2)
    assert np.array_equal(st[want == 0], raw[want == 0])


def test_emulator_settle_and_step_match_the_oracle(rb):
    from emu_lib import Emu
    from oracle_lib import Oracle
    name, b, o, e = rb
    b12 = b.set_param('NITER', 12)
    o12, e12 = Oracle(b12), Emu(b12)
    st, _ = _states(b, 2, 3001, checker=emu_checker(e))
    rng = np.random.RandomState(3)
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        o12.settle(so, 4); e12.settle(se, 4)
        assert np.abs(b.view(so[None])['q'] - b.view(se[None])['q']).max() < 1e-5
        s = so
        for k in range(3):
            a = rng.uniform(-1, 1, 7).astype(np.float32)
            s1, s2 = s.copy(), s.copy()
            o_obs, o_rew, o_done, o_info = o12.step(s1, a)
            e_obs, e_rew, e_done, e_info, _ = e12.step(s2, a)
            assert o_info[6] == e_info[6] and o_info[7] == e_info[7]
            assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4
            s = s1
