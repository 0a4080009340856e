"""DressingBaxter-v1 (BASELINE config 5) without a GPU: the model blob incl. its cloth section against the reference's numbers, the
oracle's cloth solver against properties a position-based cloth must have, and the oracle's task layer against an independent numpy
restatement of util.sleeve_on_arm_reward / dressing.py.  The cloth kernel runs one workgroup per environment and has no CPU emulator:
it is compared with the oracle on the GPU (tests/test_gpu_dressing.py).  PARITY UNPINNED vs PyBullet / the fork's cloth API."""
import numpy as np
import pytest

from assistive_gym_amd.model import compiler as L
from assistive_gym_amd.model import xform as X


@pytest.fixture(scope='module')
def dr():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('dressing_baxter')


@pytest.fixture(scope='module')
def dr_oracle(dr):
    from oracle_lib import Oracle
    return Oracle(dr)


def _states(blob, n, seed, **kw):
    from assistive_gym_amd.host.reset_dressing import make_states
    return make_states(blob, n, seed=seed, **kw)


def cloth_tables(blob):
    oc = blob.h['OFF_CLOTH']
    ci, cf = blob.i[oc:], blob.f[oc:]
    nn, nl, ncol = int(ci[L.CL['NN']]), int(ci[L.CL['NL']]), int(ci[L.CL['NCOLOR']])
    color = ci[ci[L.CL['OFF_COLOR']]:ci[L.CL['OFF_COLOR']] + ncol + 1]
    lk = ci[ci[L.CL['OFF_LINK']]:ci[L.CL['OFF_LINK']] + 2 * nl].reshape(nl, 2)[:, 0]
    rest2 = cf[ci[L.CL['OFF_LINK']]:ci[L.CL['OFF_LINK']] + 2 * nl].reshape(nl, 2)[:, 1]
    real = lk >= 0                                   # -1: an empty slot of the kernel's bank schedule (model/cloth.py bank_schedule)
    a, b = np.where(real, lk & 0xffff, -1), np.where(real, (lk >> 16) & 0xffff, -1)
    x0 = cf[ci[L.CL['OFF_X0']]:ci[L.CL['OFF_X0']] + 3 * nn].reshape(nn, 3).astype(np.float64)
    anc = ci[ci[L.CL['OFF_ANCHOR']]:ci[L.CL['OFF_ANCHOR']] + 4 * int(ci[L.CL['NANCHOR']])].reshape(-1, 4)[:, 0]
    return dict(nn=nn, nl=nl, ncol=ncol, color=color, a=a, b=b, rest2=rest2, x0=x0, anchors=anc, tri=ci[L.CL['TRI']:L.CL['TRI'] + 6],
                par=cf[ci[L.CL['OFF_PARAM']]:ci[L.CL['OFF_PARAM']] + L.CP['COUNT']])


def test_model_header_and_cloth_section(dr):
    assert dr.task_kind == L.TASK_DRESSING and dr.h['SIM_SUBSTEPS'] == 8                                    # dressing.py:184
    assert (dr.ndof, dr.nrobot, dr.nhdof, dr.nfree, dr.act_dim, dr.obs_dim) == (19, 9, 10, 0, 7, 24)       # dressing.py:9: 17 + 7
    assert [dr.robot_i(d, 'PB_INDEX') for d in range(9)] == [34, 35, 36, 37, 38, 40, 41, 49, 51]           # baxter.py:9,14
    assert [dr.robot_i(d, 'ACT') for d in range(9)] == [0, 1, 2, 3, 4, 5, 6, -1, -1]
    assert all(np.isclose(dr.robot_f(d, 'KP'), 0.01) for d in range(7))                                     # dressing.py:121
    assert [dr.robot_i(d, 'PB_INDEX', gender=0) for d in range(9, 19)] == list(range(10, 20))               # human.left_arm_joints
    coop = dr.coop()
    assert (coop.act_dim, coop.obs_dim) == (17, 24 + 28) and coop.task_i('ARM_LIMIT_ON') == 1               # dressing.py:9, human.py:136-137
    assert np.isclose(dr.task_f('ARM_LIMIT_SIGN'), 1.0)                                                     # left arm
    assert np.isclose(dr.task_f('SUCCESS_FRAC'), 0.4) and np.isclose(dr.task_f('C_D'), 0.01)                # config.ini:31,45
    assert np.allclose(dr.task_f('ARM_RADIUS', 2), [0.043, 0.0355])                                         # human_creation.py:89,140
    assert dr.param('HUMAN_GRAVITY_Z') == -1.0 and dr.param('ROBOT_GRAVITY_Z') == 0.0                       # dressing.py:179-181
    t = cloth_tables(dr)
    real = t['a'] >= 0
    assert t['nn'] == 3966 and real.sum() == 11640                                                          # SURVEY: 3,966 vertices / 7,673 faces -> 11,640 edges
    # Link schedule (model/cloth.py): 16 x K patch classes of 64 slots -- the links inside the 256-node patch of each wave of the cloth kernel --
    # then at most 16 workgroup-wide classes of the links between patches.  A class never uses a node twice (its links are relaxed in parallel);
    # the patch classes of wave w use nodes of patch w only (they are relaxed without waiting for the other waves).
    ci = dr.i[dr.h['OFF_CLOTH']:]
    K = int(ci[L.CL['NPATCH_COLOR']])
    perm = ci[ci[L.CL['OFF_PERM']]:ci[L.CL['OFF_PERM']] + 4096]
    assert 11 <= K <= 13 and 0 < t['ncol'] - 16 * K <= 16
    for c in range(t['ncol']):
        sl = slice(t['color'][c], t['color'][c + 1])
        nodes = np.concatenate([t['a'][sl], t['b'][sl]]); nodes = nodes[nodes >= 0]
        assert len(np.unique(nodes)) == len(nodes)
        if c < 16 * K:
            assert t['color'][c + 1] - t['color'][c] == 64 and set(nodes) <= set(perm[256 * (c // K):256 * (c // K + 1)])
        else:
            assert len(nodes) // 2 <= 1024
    cross = slice(t['color'][16 * K], t['color'][t['ncol']])
    assert (t['a'][cross] >= 0).sum() < 0.15 * 11640                                                         # eight of nine links lie inside a patch
    t = dict(t, a=t['a'][real], b=t['b'][real], rest2=t['rest2'][real])
    assert np.allclose(t['rest2'], np.sum((t['x0'][t['a']] - t['x0'][t['b']]) ** 2, axis=1), rtol=1e-5)
    # the four anchor nodes hang together at the attachment point (dressing.py:148,153): a check of the node order and of the load transform
    assert list(t['anchors']) == [2086, 2087, 2088, 2041]
    ap = t['x0'][t['anchors']]
    assert np.abs(ap - ap.mean(0)).max() < 0.02 and np.abs(ap.mean(0) - np.array(dr.meta['cloth_orig_pos'])).max() < 0.015
    assert list(t['tri']) == [1180, 2819, 30, 1322, 13, 696]                                                # dressing.py:156-157
    p = t['par']
    assert np.allclose([p[L.CP['KLST']], p[L.CP['KDP']], p[L.CP['KDG']], p[L.CP['KDF']], p[L.CP['PITER']], p[L.CP['MARGIN']]], [0.055, 0.01, 10, 0.39, 5, 0.04])   # dressing.py:153-154
    assert np.isclose(1.0 / p[L.CP['NODE_IM']] * t['nn'], 0.16, rtol=1e-6)                                  # total mass


def test_reset_places_the_garment_at_the_end_effector(dr, dr_oracle):
    st, cloth, infos = _states(dr, 2, 41)
    t = cloth_tables(dr)
    for i in range(2):
        assert infos[i]['toc_goals'] >= 1
        ee, _ = dr_oracle.ee_pose(st[i])
        assert np.allclose(ee, infos[i]['start_ee_pos'], atol=1e-5)
        # every node is shifted by start_ee_pos - cloth_orig_pos (dressing.py:148-153): the anchors sit at the end effector
        assert np.allclose(cloth[i, 0] - t['x0'], ee - np.array(dr.meta['cloth_orig_pos']), atol=1e-5)
        assert np.abs(cloth[i, 0, t['anchors']].mean(0) - ee).max() < 0.015 and (cloth[i, 1] == 0).all()
        v = dr.view(st[i:i + 1])
        assert np.isclose(v['task'][0, L.DR['CLOTH_GRAVITY']].view(np.float32), -9.81)                    # after the settle (none here): full gravity
        assert int(v['frozen'][0]) == 0                                                                    # the left arm stays dynamic


def test_cloth_hangs_from_its_anchors_and_keeps_its_links(dr, dr_oracle):
    """20 simulation steps of the settle (gravity -9.81 / 2): the anchors stay at the end effector, the garment falls but stays in one piece"""
    st, cloth, infos = _states(dr, 1, 43)
    s, c = st[0].copy(), cloth[0].copy()
    dr.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = np.array([-9.81 / 2], dtype=np.float32).view(np.int32)[0]
    t = cloth_tables(dr)
    z0 = c[0, :, 2].mean()
    dr_oracle.settle_cloth(s, c, 20)
    assert np.isfinite(c).all()
    ee, _ = dr_oracle.ee_pose(s)
    assert np.abs(c[0, t['anchors']].mean(0) - ee).max() < 0.02                                             # PSolve_Anchors
    assert c[0, :, 2].mean() < z0 - 0.05                                                                    # it falls ...
    real = t['a'] >= 0                               # (-1: empty slots of the link schedule)
    stretch = np.sqrt(np.sum((c[0, t['a'][real]] - c[0, t['b'][real]]) ** 2, axis=1) / t['rest2'][real])
    assert np.median(stretch) < 1.05 and np.percentile(stretch, 99) < 1.6                                   # ... as a cloth, not as loose points (kLST = 0.055 is soft)
    # the robot holds its pose during the settle (motors at their targets, no gravity on the robot)
    assert np.abs(dr.view(s.reshape(1, -1))['q'][0, :7] - dr.view(st[0:1])['q'][0, :7]).max() < 1e-3


def test_cloth_rests_on_a_capsule_at_the_margin(dr, dr_oracle):
    """the garment laid over the human's left forearm: its nodes are kept a collision margin (0.04, dressing.py:153) off the capsule and
    the contacts report forces"""
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    # anchor hardness 0: PSolve_Anchors then only cancels the anchored nodes' displacement, i.e. pins them where they are, so that the
    # garment can be laid over the arm without being dragged back to the end effector
    w = dr.words.copy()
    oc = dr.h['OFF_CLOTH']
    w.view(np.float32)[oc + int(dr.i[oc + L.CL['OFF_PARAM']]) + L.CP['KAHR']] = 0.0
    dr, dr_oracle = ModelBlob(w, dr.meta), None
    dr_oracle = Oracle(dr)
    st, cloth, infos = _states(dr, 1, 47)
    s, c = st[0].copy(), cloth[0].copy()
    pos, rot = dr_oracle.fk(s)
    links = dr.task_i_n('OBS_LINK', 3)
    elbow, wrist = pos[links[1]], pos[links[2]]
    t = cloth_tables(dr)
    centre_node = int(np.argmin(np.linalg.norm(t['x0'] - t['x0'].mean(0), axis=1)))
    gender = int(dr.view(s.reshape(1, -1))['gender'][0])
    from assistive_gym_amd.model.human import HumanModel
    r = HumanModel('male' if gender == 0 else 'female', cloth=True).dims['forearm'][0]
    # shift the whole garment so that its central node lies 1 cm outside the forearm's margin shell, right above the axis (the rest of it
    # then cuts through the torso and the wheelchair: those nodes are pushed out during the step)
    c[0] += (0.5 * (elbow + wrist) + np.array([0, 0, r + 0.05]) - c[0, centre_node]).astype(np.float32)

    obs, rew, done, info = dr_oracle.step_cloth(s, c, np.zeros(7, dtype=np.float32))
    con = dr_oracle.cloth_contacts()
    pos2, rot2 = dr_oracle.fk(s)
    # clearance of every node to every capsule / sphere of the human's moving left arm
    x = c[0].astype(np.float64)
    clear = np.full(len(x), np.inf)
    rg = dr.meta['ranges']['human_male' if gender == 0 else 'human_female']
    for ci in range(*rg):
        col = dr.collider(ci)
        if col['body'] >= 100 or len(col['verts']) > 2:
            continue
        pw = pos2[col['body']] + col['verts'] @ rot2[col['body']].T
        e, w_ = pw[0], pw[-1]
        den = np.dot(w_ - e, w_ - e)
        tt = np.clip(((x - e) @ (w_ - e)) / den, 0, 1) if den > 0 else np.zeros(len(x))
        clear = np.minimum(clear, np.linalg.norm(x - (e + tt[:, None] * (w_ - e)), axis=1) - col['radius'])
    # after 40 substeps no node sits deeper in a margin shell than one substep's fall, and part of the garment rests ON the shell
    assert (clear < 0.04 - 0.012).mean() < 0.002, (clear < 0.028).sum()
    assert ((clear > 0.03) & (clear < 0.05)).sum() >= 10
    assert len(con) >= 10 and (np.linalg.norm(con[:, 3:], axis=1) > 0).any()
    assert info[3] >= 0 and np.isclose(obs[23], info[3])       # cloth_force_sum reaches the observation (dressing.py:96)


def sleeve_reward_np(tri1, tri2, shoulder, elbow, wrist, rad):
    """independent numpy restatement of Util.sleeve_on_arm_reward + the reward branches of DressingEnv.step (util.py:134-202, dressing.py:49-59)"""
    def unit(v):
        return v / np.linalg.norm(v)
    hand_end = wrist + unit(wrist - elbow) * rad * 2
    elbow_end = elbow + unit(elbow - wrist) * rad
    shoulder_end = shoulder + unit(shoulder - elbow) * rad
    pts = np.concatenate([tri1, tri2])

    def around(axis_from, axis_to, origin):
        n = unit(axis_to - axis_from)
        t = unit(np.cross([1, 1, 0], n))
        b = unit(np.cross(t, n))
        tp, bp = (pts - origin) @ t, (pts - origin) @ b
        return (tp > 0).any() and (tp < 0).any() and (bp > 0).any() and (bp < 0).any()

    def vol(a, b, c, d):
        return np.dot(np.cross(b - a, c - a), d - a) / 6.0

    def hits(p, q0, q1):
        if np.sign(vol(q0, *p)) != np.sign(vol(q1, *p)):
            return np.sign(vol(q0, q1, p[0], p[1])) == np.sign(vol(q0, q1, p[1], p[2])) == np.sign(vol(q0, q1, p[2], p[0]))
        return False
    fore = around(elbow_end, hand_end, hand_end) and (hits(tri1, hand_end, elbow_end) or hits(tri2, hand_end, elbow_end))
    upper = around(shoulder_end, elbow_end, shoulder_end) and (hits(tri1, elbow_end, shoulder_end) or hits(tri2, elbow_end, shoulder_end))
    centre = pts.mean(0)
    d_hand = np.linalg.norm(hand_end - centre)
    d_upper = np.linalg.norm(centre - elbow)
    forearm_length, upperarm_length = np.linalg.norm(hand_end - elbow_end), np.linalg.norm(elbow - shoulder)
    if upper:
        return forearm_length + (d_upper if d_upper < upperarm_length else 0.0), fore, upper
    if fore and d_hand < forearm_length:
        return d_hand, fore, upper
    return -d_hand, fore, upper


@pytest.mark.parametrize('where', ['away', 'forearm', 'upperarm'])
def test_dressing_reward_matches_numpy_restatement(dr, dr_oracle, where):
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    # a frozen garment (no solver iterations, no gravity): the six sleeve vertices stay where the test puts them
    w = dr.words.copy()
    oc = dr.h['OFF_CLOTH']
    w.view(np.float32)[oc + int(dr.i[oc + L.CL['OFF_PARAM']]) + L.CP['PITER']] = 0.0
    dr = ModelBlob(w, dr.meta)
    dr_oracle = Oracle(dr)
    st, cloth, infos = _states(dr, 1, 53)
    s, c = st[0].copy(), cloth[0].copy()
    dr.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = 0
    t = cloth_tables(dr)
    links = dr.task_i_n('OBS_LINK', 3)
    c[0] += np.array([0, 0, 5.0], dtype=np.float32)          # the garment out of the way
    if where != 'away':
        pos, _ = dr_oracle.fk(s)
        sh, el, wr = pos[links[0]], pos[links[1]], pos[links[2]]
        p0, p1 = (el, wr) if where == 'forearm' else (sh, el)
        axis = (p1 - p0) / np.linalg.norm(p1 - p0)
        u = np.cross(axis, [0, 0, 1.0]); u /= np.linalg.norm(u)
        w = np.cross(axis, u)
        mid = p0 + 0.5 * (p1 - p0)
        ring = [mid + 0.15 * (np.cos(a) * u + np.sin(a) * w) for a in np.deg2rad([0, 120, 240])]
        ring2 = [mid + 0.02 * axis + 0.15 * (np.cos(a) * u + np.sin(a) * w) for a in np.deg2rad([60, 180, 300])]
        for n, p in zip(t['tri'], ring + ring2):               # two triangles around the limb, pierced by its axis
            c[0, n] = p
            c[1, n] = 0
    act = np.array([0.3, -0.2, 0.1, 0, 0, 0, 0.5], dtype=np.float32)
    obs, rew, done, info = dr_oracle.step_cloth(s, c, act)
    pos, _ = dr_oracle.fk(s)
    gender = int(dr.view(s.reshape(1, -1))['gender'][0])
    want, fore, upper = sleeve_reward_np(c[0, t['tri'][:3]].astype(np.float64), c[0, t['tri'][3:]].astype(np.float64), pos[links[0]], pos[links[1]], pos[links[2]],
                                         dr.task_f('ARM_RADIUS', 2)[gender])
    assert {'away': not fore and not upper, 'forearm': fore and not upper, 'upperarm': bool(upper)}[where]
    assert np.isclose(info[4], want, atol=2e-4), (info[4], want)            # AGX_INFO_FOOD_REWARD carries reward_dressing
    ee_speed = -(rew - want * 1.0 + 0.01 * np.linalg.norm(act) + 0.01 * info[3]) / 0.25     # reward = dressing - 0.01 |a| + C_v (-v) + C_d (-cloth forces)
    assert 0 <= ee_speed < 1.0
    assert obs.shape == (24,) and np.isclose(obs[23], info[3]) and not done
    v = dr.view(s.reshape(1, -1))
    assert np.isclose(v['task'][0, L.DR['BEST']].view(np.float32), max(want, 0.0), atol=2e-4)      # self.task_success starts at 0 (dressing.py:62-63)
    assert info[1] == float(max(want, 0.0) >= 0.4)


def test_observation_layout(dr, dr_oracle):
    st, cloth, infos = _states(dr, 1, 59)
    s = st[0].copy()
    obs = dr_oracle.observe(s)
    ee, eq = dr_oracle.ee_pose(s)
    v = dr.view(s.reshape(1, -1))
    bp, bq = v['base'][0, :3].astype(np.float64), v['base'][0, 3:].astype(np.float64)
    ip, iq = X.invert(bp, bq)
    pr, qr = X.compose(ip, iq, ee, eq)
    assert np.allclose(obs[0:3], pr, atol=1e-5) and (np.allclose(obs[3:7], qr, atol=1e-5) or np.allclose(obs[3:7], -qr, atol=1e-5))      # dressing.py:79-80
    q = v['q'][0, :7].astype(np.float64)
    assert np.allclose(obs[7:14], (q + np.pi) % (2 * np.pi) - np.pi, atol=1e-5)                                                          # :83
    pos, _ = dr_oracle.fk(s)
    for k, link in enumerate(dr.task_i_n('OBS_LINK', 3)):
        assert np.allclose(obs[14 + 3 * k:17 + 3 * k], X.apply(ip, iq, pos[link][None])[0], atol=1e-5)                                   # :86-91
    assert obs[23] == 0                                                                                                                   # cloth_force_sum after reset


def test_internal_substeps_and_hooks(dr, dr_oracle):
    """numSubSteps = 8 (dressing.py:184): one env step = 5 x 8 internal substeps of 2.5 ms; the joint-limit reset of the human runs once per
    stepSimulation call.  The rigid scene stepped without a garment attached gives the same joint angles as with one (one-way coupling)."""
    st, cloth, infos = _states(dr, 1, 61, impairment='tremor')
    act = np.array([1, -1, 0.5, 0, 0, 1, -1], dtype=np.float32)
    s1, s2, c = st[0].copy(), st[0].copy(), cloth[0].copy()
    o1 = dr_oracle.step(s1, act)
    o2 = dr_oracle.step_cloth(s2, c, act)
    v1, v2 = dr.view(s1.reshape(1, -1)), dr.view(s2.reshape(1, -1))
    assert np.array_equal(v1['q'], v2['q']) and np.array_equal(v1['qd'], v2['qd'])
    dq = np.abs(v1['q'][0, :7] - dr.view(st[0:1])['q'][0, :7])
    assert dq.max() > 0.01                                       # the arm follows its targets
    assert int(v1['iteration'][0]) == 1


def test_emulator_matches_oracle_on_the_rigid_scene(dr, dr_oracle):
    """the device code of the `dressing` variant (40 internal substeps, hooks after every 8th, the row-space solve, the dressing task layer
    without a garment attached) on the CPU wave emulator"""
    from emu_lib import Emu
    e = Emu(dr)
    st, cloth, infos = _states(dr, 2, 67, impairment='tremor')
    rng = np.random.RandomState(2)
    for i in range(2):
        so, se = st[i].copy(), st[i].copy()
        for k in range(2):
            a = rng.uniform(-1, 1, 7).astype(np.float32)
            o_obs, o_rew, o_done, o_info = dr_oracle.step(so, a)
            e_obs, e_rew, e_done, e_info, _ = e.step(se, a)
            assert np.abs(o_obs - e_obs).max() < 1e-4 and abs(o_rew - e_rew) < 1e-4 and o_done == e_done
            assert np.abs(dr.view(so.reshape(1, -1))['q'] - dr.view(se.reshape(1, -1))['q']).max() < 2e-5
