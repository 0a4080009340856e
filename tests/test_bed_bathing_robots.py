"""BedBathing<Robot>-v1 for Jaco, Panda (on their nightstand), PR2 and Baxter (more robots on the bed-bathing kernels)
without a GPU: blobs against the reference's robot tables, the host reset, the collision pass and the kernel variant on the wave emulator
against the oracle incl. a wiping contact.  The rag-doll settle is replaced by the rigid 'drop' stand-in here (no GPU); the GPU tests use
the real one.  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X
from bed_util import arm_points, move_pad_to
from test_scratch_itch_robots import emu_checker, flags_from_oracle

from conftest import full

ROBOTS = ['jaco', pytest.param('panda', marks=full), 'pr2', pytest.param('baxter', marks=full)]


@pytest.fixture(scope='module', params=ROBOTS)
def rb(request):
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    from oracle_lib import Oracle
    b = ModelBlob.load('bed_bathing_' + request.param)
    return request.param, b, Oracle(b), Emu(b)


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset_bed import make_states
    return make_states(blob, n, seed=seed, **kw)


def test_model_tables(rb):
    from assistive_gym_amd.model import compiler as L
    name, b, o, e = rb
    T = L.robot_table('bed_bathing', name)
    assert b.task_kind == L.TASK_BED_BATHING and (b.act_dim, b.obs_dim, b.nhdof) == (7, 24, 10)              # bed_bathing.py:10: 17 + 7
    arm_dofs = sorted((d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0), key=lambda d: b.robot_i(d, 'ACT'))
    assert [b.robot_i(d, 'PB_INDEX') for d in arm_dofs] == T['arm']
    grip_dofs = [d for d in range(b.nrobot) if b.robot_i(d, 'PB_INDEX') in T['grip']]
    assert np.allclose([b.robot_f(d, 'QT0') for d in grip_dofs], T['gripper_target'])                        # gripper_pos['bed_bathing']
    assert b.nrobot == len(T['arm']) + len(T['grip'])                                                        # every other joint is static geometry
    assert b.task_i_n('NT', 4) == [81, 48, 56, 35] and np.isclose(b.param('HUMAN_GRAVITY_Z'), -1.0)
    r = b.meta['ranges']
    base_links = [b.collider(c)['link'] for c in range(*r['robot_base'])]
    if T['wheelchair_mounted']:     # the nightstand travels with the base: a hull about 0.55 m tall whose top is just under the arm's base
        ns = b.collider(r['robot_base'][1] - 1)['verts']
        assert 0.4 < np.ptp(ns[:, 2]) < 0.75 and abs(ns[:, 2].max()) < 0.12 and np.linalg.norm(ns[:, :2].mean(0)) < 0.25
    else:
        assert len(base_links) >= 1
    c = b.coop()
    assert (c.act_dim, c.obs_dim) == (17, 24 + 28)


def test_reset_and_collision_pass(rb):
    name, b, o, e = rb
    n = 8
    raw, infos = _states(b, n, 2001)
    want_q = X.quat_from_rpy(b.meta['ee_rpy'])
    for i in range(n):
        p, q = o.ee_pose(raw[i])
        assert np.linalg.norm(p - infos[i]['target_ee_pos']) < 0.031                                         # robot.py:97 success_threshold
        assert min(np.linalg.norm(q - want_q), np.linalg.norm(q + want_q)) < 0.031
        assert infos[i]['toc_goals'] >= 2                                                                    # the start pose and at least one of shoulder / elbow / wrist
    got = emu_checker(e)(raw)
    want = np.array([flags_from_oracle(b, o, s) for s in raw])
    assert np.array_equal(got, want)


def test_emulator_matches_oracle_in_free_space_and_wiping(rb):
    name, b, o, e = rb
    st, infos = _states(b, 2, 4001)
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        for k in range(3):
            a = np.random.RandomState(10 * i + k).uniform(-1, 1, 7).astype(np.float32)
            oo, orr, od, oi = o.step(so, a)
            eo, er, ed, ei, _ = e.step(se, a)
            assert oi[6] == ei[6] and oi[7] == ei[7]
            assert np.abs(oo - eo).max() < 2e-5 and abs(orr - er) < 2e-5
    # the wiping pad pressed 4 mm into the top of the forearm, the arm abducted (as tests/test_bed_bathing.py wiping_state)
    st, infos = _states(b, 1, 4101, human_q_override={3: np.deg2rad(70)})
    s = st[0].copy()
    sh, el, wr, _ = arm_points(b, o, s)
    c = b.collider([k for k in range(*b.meta['ranges']['human_male' if infos[0]['gender'] == 'male' else 'human_female']) if b.collider(k)['link'] == 7][0])
    move_pad_to(b, s, el + 0.5 * (wr - el) + np.array([0, 0, c['radius'] + 0.0025 - 0.004]))
    so, se = s.copy(), s.copy()
    wiped = 0
    for k in range(3):
        a = (np.random.RandomState(k).uniform(-1, 1, 7) * 0.2).astype(np.float32)
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6] and oi[4] == ei[4]
        assert np.abs(oo[:-1] - eo[:-1]).max() < 1e-4 and abs(oo[-1] - eo[-1]) <= 1e-3 * max(1.0, abs(oo[-1]))
        for col in (0, 2, 3):
            assert abs(oi[col] - ei[col]) <= 1e-3 * max(1.0, abs(oi[col]))
        wiped += int(oi[4])
    assert wiped >= 1, 'the pad wipes targets off the forearm'
