"""The rag-doll settle of BedBathingEnv.reset (bed_bathing.py:119-137) without a GPU: the bed_settle model (the whole human as
one floating 47-DoF articulated body), the oracle's run of it, and the device code (bed_settle kernel variant on the CPU wave
emulator) against the oracle.  PARITY UNPINNED vs PyBullet (oracle/agx_oracle.h)."""
import numpy as np
import pytest

from assistive_gym_amd.model import xform as X


@pytest.fixture(scope='module')
def sb():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('bed_settle')


@pytest.fixture(scope='module')
def bb():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('bed_bathing_sawyer')


@pytest.fixture(scope='module')
def s_oracle(sb):
    from oracle_lib import Oracle
    return Oracle(sb)


def posed(sb, bb, seed, **kw):
    from assistive_gym_amd.host import reset_bed as rb
    rs = rb.BedBathingSawyerReset(bb)
    pre = rs.pre_settle(np.random.RandomState(seed), **kw)
    ss = sb.new_state(1)
    rb.settle_record(sb, ss[0:1], pre['gender'], pre['limit_scale'], pre['base_pos'], pre['base_rpy'], pre['hq'], pre['plane_friction'])
    return ss[0], pre


def test_settle_model_tables(sb):
    from assistive_gym_amd.model import compiler as L
    from assistive_gym_amd.model.human import HumanModel
    assert (sb.ndof, sb.nrobot, sb.nhdof, sb.nfree, sb.act_dim) == (47, 0, 47, 0, 0)
    assert [sb.robot_i(d, 'JTYPE') for d in range(6)] == [1, 1, 1, 0, 0, 0]
    assert [sb.robot_i(d, 'PARENT') for d in range(1, 6)] == [0, 1, 2, 3, 4]
    assert all(sb.robot_f(d, 'MAXF') == 0 for d in range(47))                                   # setup_joints(..., reactive_force=None): no motors
    assert all(sb.robot_i(d, 'KIND') & 4 for d in range(47))                                    # limits act as constraints only during the settle
    hm = HumanModel('male')
    mass = sum(sb.robot_f(d, 'MASS') for d in range(47))
    assert np.isclose(mass, hm.mass.sum() + 0.1 * hm.total_mass, rtol=1e-6)                     # human_creation.py:280
    joints = sb.meta['settle_joints']
    assert len(joints) == 41 and 24 not in joints
    # the impairment scales the limits of the arm and head joints only (human_creation.py passes limit_scale to those)
    scaled = [joints[d - 6] for d in range(6, 47) if sb.robot_i(d, 'KIND') == 5]
    half = HumanModel('male', 0.5)
    assert scaled == [j for j in joints if not np.isclose(half.lower[j], hm.lower[j]) or not np.isclose(half.upper[j], hm.upper[j])]
    assert sb.param('HUMAN_GRAVITY_Z') == -1.0                                                   # bed_bathing.py:123
    r = sb.meta['ranges']
    assert r['human_male_rarm'][1] - r['human_male_rarm'][0] == 3                                # upper arm, forearm, hand


def test_settle_record_matches_human_fk(sb, bb, s_oracle):
    s, pre = posed(sb, bb, 5)
    pos, rot = s_oracle.fk(s)
    hp, hq = pre['hm'].fk(pre['base_pos'], pre['base_quat'], pre['hq'])
    assert np.allclose(pos[5], pre['base_pos'], atol=1e-6) and np.allclose(rot[5], X.quat_to_mat(pre['base_quat']), atol=1e-6)
    for k, j in enumerate(sb.meta['settle_joints']):
        assert np.allclose(pos[6 + k], hp[j], atol=1e-6) and np.allclose(rot[6 + k], X.quat_to_mat(hq[j]), atol=1e-6)


def test_free_fall_before_contact(sb, bb, s_oracle):
    """no bed contact during the first 10 simulation steps: the base falls under g = -1 and the joints that no self-collision pushes
    (head, waist, legs; the arms start inside the torso's capsules and are pushed out) keep their angles: no motors, and gravity
    accelerates every link alike"""
    s, pre = posed(sb, bb, 7)
    s0 = s.copy()
    s_oracle.settle(s, 10)
    v, v0 = sb.view(s.reshape(1, -1)), sb.view(s0.reshape(1, -1))
    t = 10 * 0.02
    assert v['q'][0, 2] < v0['q'][0, 2] - 0.4 * t * t and v['q'][0, 2] > v0['q'][0, 2] - 0.5 * t * t * 1.1 - 1e-3
    assert np.abs(v['q'][0, :2] - v0['q'][0, :2]).max() < 1e-3
    free = [6 + k for k, j in enumerate(sb.meta['settle_joints']) if j >= 20]
    assert np.abs(v['q'][0, free] - v0['q'][0, free]).max() < 2e-2


def test_oracle_settle_rests_on_the_bed(sb, bb, s_oracle):
    from assistive_gym_amd.host import reset_bed as rb
    for seed, kw in ((5, {}), (6, dict(gender='female', impairment='limits'))):
        s, pre = posed(sb, bb, seed, **kw)
        s_oracle.settle(s, 100)
        v = sb.view(s.reshape(1, -1))
        assert np.abs(v['qd'][0]).max() < 2.0 and np.abs(v['qd'][0, :3]).max() < 0.05           # at rest up to limb wobble
        bp, bq, hq = rb.settled_pose(sb, s.reshape(1, -1), pre['hm'])
        hm = pre['hm']
        assert np.max(np.maximum(hm.lower - hq, hq - hm.upper)) < 5e-3                           # joint limits held as constraints
        hp, _ = hm.fk(bp, bq, hq)
        # lying on the mattress: the chest a chest-radius above it, every link frame within the mattress footprint and above it
        top = rb.BedBathingSawyerReset(bb)._bed_top(bp[0], bp[1])
        assert abs(bp[2] - (top + hm.dims['chest'][0])) < 0.04                                 # the mattress hulls are not flat: a loose bound
        con = s_oracle.collide(s)
        bed0 = sb.meta['ranges']['bed'][0]
        on_bed = con[con[:, 1] >= bed0]
        assert len(on_bed) >= 8 and on_bed[:, 11].min() > -5e-3 and (on_bed[:, 10] > 0.8).all()  # resting contacts: shallow, normals up
        assert hp[:, 2].min() > top - 0.08 and hp[:, 2].max() < top + 0.3                       # the right arm lies along the mattress edge
        assert abs(X.quat_to_mat(bq)[1, 2]) > 0.95                                               # still on the back: chest z axis along world -y... (rpy[0] ~ -pi/2)


def test_emulator_matches_oracle_through_the_impact(sb, bb, s_oracle):
    """the device code of the bed_settle variant on the CPU wave emulator: free fall, first contacts and the early settle"""
    from emu_lib import Emu
    e = Emu(sb)
    s, pre = posed(sb, bb, 5)
    so, se = s.copy(), s.copy()
    s_oracle.settle(so, 18)                                                                      # contact begins around step 20
    e.settle(se, 18)
    assert np.abs(so[:47] - se[:47]).max() < 1e-4 and np.abs(so[47:94] - se[47:94]).max() < 2e-3   # 18 steps of f32 against f64, the arms in self-contact
    for _ in range(6):
        se = so.copy()                                                                           # single steps from the oracle's trajectory: no chaotic drift
        s_oracle.settle(so, 1)
        e.settle(se, 1)
        v, w = sb.view(so.reshape(1, -1)), sb.view(se.reshape(1, -1))
        assert np.abs(v['q'][0] - w['q'][0]).max() < 2e-5
        assert np.abs(v['qd'][0] - w['qd'][0]).max() < 2e-3


def test_make_states_with_a_settler(sb, bb, s_oracle):
    """host reset around the rag-doll settle (here run by the oracle; the product runs it on the device, RagdollSettler)"""
    from assistive_gym_amd.host import reset_bed as rb

    class OracleSettler:
        blob = sb

        def __call__(self, states):
            out = states.copy()
            for r in out:
                s_oracle.settle(r, 100)
            return out
    st, infos = rb.make_states(bb, 2, seed=1001, settler=OracleSettler())
    st_drop, infos_drop = rb.make_states(bb, 2, seed=1001)
    for i in range(2):
        v = bb.view(st[i:i + 1])
        assert infos[i]['gender'] == infos_drop[i]['gender'] and infos[i]['impairment'] == infos_drop[i]['impairment']
        assert infos[i]['toc_goals'] >= 1
        bp, bq = infos[i]['human_base']
        assert 0 < infos_drop[i]['human_base'][0][2] - bp[2] < 0.15                               # the rigid drop stops at the first (bounding-box) touch, the rag doll sinks into the mattress' shape
        assert np.abs(infos[i]['human_q'] - infos_drop[i]['human_q']).max() > 0.05                # but the limbs came to rest, not frozen in the air
        hp, hq = infos[i]['human_q'], None
        # the stepper's record is consistent with the settled pose
        pos, quat = rb.BedBathingSawyerReset(bb)._human(infos[i]['gender'], infos[i]['limit_scale']).fk(bp, bq, infos[i]['human_q'])
        for k, link in enumerate(bb.meta['human_bodies']):
            want = bp if link < 0 else pos[link]
            assert np.allclose(v['human'][0, k, :3], want, atol=1e-5)
        assert np.allclose(v['q'][0, bb.nrobot:], [infos[i]['human_q'][j] for j in bb.meta['human_dynamic_joints']], atol=1e-6)
