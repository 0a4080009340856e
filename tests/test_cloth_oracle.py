"""The oracle's cloth substep (oracle/agx_oracle.c::cloth_substep) against an independent, deliberately plain numpy restatement of the
same position-based step, on a small synthetic garment (a 6 x 5 grid of nodes hung by two anchors) spliced into the DressingBaxter blob
in place of the hospital gown: gravity, the one-sided aerodynamic drag with its clamp, the anchor update, contacts with one capsule of
the human (margin shell, friction state), links class by class, the velocity update -- and the contact report.  What both sides share is
the blob (links, classes, face table, parameters); the arithmetic is written twice."""
import os

import numpy as np
import pytest

from assistive_gym_amd.model import compiler as L


def _grid_obj(path, nx=6, ny=5, h=0.03):
    with open(path, 'w') as f:
        for j in range(ny):
            for i in range(nx):
                f.write('v %f %f %f\n' % (i * h, j * h, 0.002 * ((i * 7 + j * 3) % 5)))      # slightly crumpled: normals are not all alike
        for j in range(ny - 1):
            for i in range(nx - 1):
                a, b, c, d = j * nx + i + 1, j * nx + i + 2, (j + 1) * nx + i + 2, (j + 1) * nx + i + 1
                f.write('f %d//%d %d//%d %d//%d\n' % (a, a, b, b, c, c))
                f.write('f %d//%d %d//%d %d//%d\n' % (a, a, c, c, d, d))


@pytest.fixture(scope='module')
def small(tmp_path_factory):
    """(blob with the synthetic garment, oracle, collider index of the human's left forearm capsule)"""
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.model.cloth import compile_cloth
    from oracle_lib import Oracle
    dr = ModelBlob.load('dressing_baxter')
    obj = str(tmp_path_factory.mktemp('cloth') / 'grid.obj')
    _grid_obj(obj)
    colliders = [dr.collider(c) for c in range(dr.h['NCOLL'])]
    r = dr.meta['ranges']['human_male']
    # the male's forearm capsule: a two-vertex core on a moving link of the left arm
    cands = [c for c in range(*r) if len(colliders[c]['verts']) == 2 and colliders[c]['link'] == 17]      # human.left_elbow's link: the forearm
    assert cands, 'forearm capsule not found'
    shape = cands[0]
    sec, meta = compile_cloth(obj, 1.0, [0, 0, 0], [0, 0, 0], [0, 5], [0.0, 0.0, 0.0], [0, 1, 2], [3, 4, 5],
                              dict(KLST=0.055, KDP=0.01, KDG=10.0, KDF=0.39, KCHR=1.0, KKHR=1.0, KAHR=1.0, PITER=5, MARGIN=0.04, MASS=0.16 * 30 / 3966,
                                   AIR_DENSITY=1.2, FORCE_SCALE=10.0, FORCE_MAX=20.0, EE_BELOW=0.05), colliders, [shape], gender_of=lambda ci: 1)
    oc = dr.h['OFF_CLOTH']
    w = np.concatenate([dr.words[:oc], sec])
    w[L.H['NWORDS']] = len(w)
    blob = ModelBlob(w, dr.meta)
    return blob, Oracle(blob), shape


def tables(blob):
    oc = blob.h['OFF_CLOTH']
    ci, cf = blob.i[oc:], blob.f[oc:]
    nn, nl, ncol = int(ci[L.CL['NN']]), int(ci[L.CL['NL']]), int(ci[L.CL['NCOLOR']])
    lk = ci[ci[L.CL['OFF_LINK']]:ci[L.CL['OFF_LINK']] + 2 * nl].reshape(nl, 2)[:, 0]
    node = ci[ci[L.CL['OFF_NODE']]:ci[L.CL['OFF_NODE']] + 2 * (nn + 1)].reshape(nn + 1, 2)[:, 0]
    area = cf[ci[L.CL['OFF_NODE']]:ci[L.CL['OFF_NODE']] + 2 * (nn + 1)].reshape(nn + 1, 2)[:nn, 1].astype(np.float64)
    nface = int(node[nn])
    face = ci[ci[L.CL['OFF_FACE']]:ci[L.CL['OFF_FACE']] + nface]
    anc = ci[ci[L.CL['OFF_ANCHOR']]:ci[L.CL['OFF_ANCHOR']] + 4 * int(ci[L.CL['NANCHOR']])].reshape(-1, 4)
    ancf = cf[ci[L.CL['OFF_ANCHOR']]:ci[L.CL['OFF_ANCHOR']] + 4 * int(ci[L.CL['NANCHOR']])].reshape(-1, 4)[:, 1:].astype(np.float64)
    par = cf[ci[L.CL['OFF_PARAM']]:ci[L.CL['OFF_PARAM']] + L.CP['COUNT']].astype(np.float64)
    real = lk >= 0                                   # -1: an empty slot of the kernel's bank schedule
    rest2 = cf[ci[L.CL['OFF_LINK']]:ci[L.CL['OFF_LINK']] + 2 * nl].reshape(nl, 2)[:, 1].astype(np.float64)
    return dict(nn=nn, a=(lk & 0xffff)[real], b=((lk >> 16) & 0xffff)[real], rest2=rest2[real],
                node=node, face=face, area=area, anchors=anc[:, 0], anchor_off=ancf, par=par,
                x0=cf[ci[L.CL['OFF_X0']]:ci[L.CL['OFF_X0']] + 3 * nn].reshape(nn, 3).astype(np.float64))


def numpy_substep(t, x, v, grav, dt, anchor, capsule=None, friction=0.5):
    """one internal substep, written independently of the C code; capsule = (p0, p1, radius) in world coordinates or None"""
    P = t['par']
    kLST, kDP, kDG, kDF, kAHR, mrg, im, rho = (P[L.CP[k]] for k in ('KLST', 'KDP', 'KDG', 'KDF', 'KAHR', 'MARGIN', 'NODE_IM', 'AIR_DENSITY'))
    nn = t['nn']
    x, v = x.copy(), v.copy()
    # normals from the incident faces (area weighted), then gravity and the one-sided drag
    for i in range(nn):
        n = np.zeros(3)
        for e in range(t['node'][i], t['node'][i + 1]):
            j, k = t['face'][e] & 0xffff, (t['face'][e] >> 16) & 0xffff
            n += np.cross(x[j] - x[i], x[k] - x[i])
        ln = np.linalg.norm(n)
        if ln > 1.1920929e-7:
            n /= ln
        v[i, 2] += grav * dt
        s2 = v[i] @ v[i]
        if s2 > 1.1920929e-7 and v[i] @ n > 0:
            f = t['area'][i] * (v[i] @ n) * s2 / 2 * rho * kDG            # magnitude of the drag, against the velocity
            if (f * dt * im) ** 2 > s2:
                v[i] = 0
            else:
                v[i] = v[i] - v[i] / np.sqrt(s2) * f * dt * im
    q = x.copy()
    x = q + v * dt
    contacts = {}
    if capsule is not None:
        p0, p1, rad = capsule
        for i in range(nn):
            if i in t['anchors']:
                continue
            u = np.clip((x[i] - p0) @ (p1 - p0) / ((p1 - p0) @ (p1 - p0)), 0, 1)
            d = x[i] - (p0 + u * (p1 - p0))
            dist = np.linalg.norm(d) - rad - mrg
            if dist < 0:
                n = d / np.linalg.norm(d)
                vr = x[i] - q[i]
                dn = vr @ n
                fv = vr - n * dn
                fc = kDF * friction
                contacts[i] = dict(n=n, off=-(n @ x[i]) + dist, c3=0.0 if fv @ fv < (dn * fc) ** 2 else 1 - fc, imp=np.zeros(3))
    for it in range(int(P[L.CP['PITER']])):
        for a, i in enumerate(t['anchors']):
            x[i] = x[i] - (x[i] - q[i]) + (anchor + t['anchor_off'][a] - x[i]) * kAHR
        for i, c in contacts.items():
            vr = x[i] - q[i]
            dn = vr @ c['n']
            if dn <= 1.1920929e-7:
                dp = min(x[i] @ c['n'] + c['off'], mrg)
                corr = vr - (vr - c['n'] * dn) * c['c3'] + c['n'] * dp
                x[i] = x[i] - corr
                c['imp'] += corr / (dt * im)
        for l in range(len(t['a'])):          # the blob lists the links class by class; within a class the order is immaterial
            a, b = t['a'][l], t['b'][l]
            d = x[b] - x[a]
            ln = d @ d
            if t['rest2'][l] + ln > 1.1920929e-7:
                k = (t['rest2'][l] - ln) / (t['rest2'][l] + ln) * kLST * 0.5
                x[a] -= d * k
                x[b] += d * k
    v = (x - q) / dt * (1 - kDP)
    return x, v, {i: (x[i].copy(), c['imp'] / dt) for i, c in contacts.items()}


def _record(blob, oracle, seed=71):
    from assistive_gym_amd.host.reset_dressing import DressingBaxterReset
    from assistive_gym_amd.blob import ModelBlob
    full = ModelBlob.load('dressing_baxter')
    st = full.new_state(1)
    cl = np.zeros((2, 3966, 3), dtype=np.float32)
    DressingBaxterReset(full).sample(np.random.RandomState(seed), st, cl, env_seed=seed, gender='male', impairment='none')
    return st[0]


def test_free_hanging_patch_matches_numpy(small):
    blob, oracle, shape = small
    t = tables(blob)
    s = _record(blob, oracle)
    ee, _ = oracle.ee_pose(s)
    rng = np.random.RandomState(3)
    x = t['x0'] + ee + np.array([0, 0, 0.3])                       # the patch well above everything: no contact
    v = rng.uniform(-0.5, 0.5, x.shape)
    v[:, 2] -= 0.5
    cloth = np.stack([x, v]).astype(np.float32)
    blob.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = np.array([-9.81], dtype=np.float32).view(np.int32)[0]
    xr, vr = cloth[0].astype(np.float64), cloth[1].astype(np.float64)
    dt = 0.02 / 8
    for k in range(8):                                             # one stepSimulation = 8 internal substeps, the attachment fixed at the end effector
        xr, vr, _ = numpy_substep(t, xr, vr, -9.81, dt, ee)
    oracle.settle_cloth(s, cloth, 1)
    assert np.abs(cloth[0] - xr).max() < 2e-6 and np.abs(cloth[1] - vr).max() < 2e-4


def test_patch_on_the_forearm_matches_numpy(small):
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    blob, oracle, shape = small
    # anchor hardness 0 pins the two anchored nodes where they are instead of dragging the patch to the end effector
    w = blob.words.copy()
    oc = blob.h['OFF_CLOTH']
    w.view(np.float32)[oc + int(blob.i[oc + L.CL['OFF_PARAM']]) + L.CP['KAHR']] = 0.0
    blob = ModelBlob(w, blob.meta)
    oracle = Oracle(blob)
    t = tables(blob)
    s = _record(blob, oracle)
    pos, rot = oracle.fk(s)
    col = blob.collider(shape)
    pw = pos[col['body']] + col['verts'] @ rot[col['body']].T
    ee, _ = oracle.ee_pose(s)
    mid = 0.5 * (pw[0] + pw[1])
    x = t['x0'] - t['x0'].mean(0) + mid + np.array([0, 0, col['radius'] + 0.035])     # the patch inside the margin shell above the capsule
    v = np.zeros_like(x)
    v[:, 0] = 0.05                                                                   # sliding: exercises the friction state
    cloth = np.stack([x, v]).astype(np.float32)
    blob.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = np.array([-9.81], dtype=np.float32).view(np.int32)[0]
    xr, vr = cloth[0].astype(np.float64), cloth[1].astype(np.float64)
    dt = 0.02 / 8
    s_before = s.copy()
    oracle.step_cloth(s, cloth, np.zeros(7, dtype=np.float32))                       # 40 substeps; the arm barely moves (held by its PD)
    con = oracle.cloth_contacts()
    # replay with the capsule where the oracle's arm was at the START of each substep: re-run the rigid scene alone, substep by substep
    from assistive_gym_amd.blob import ModelBlob
    s2 = s_before.copy()
    rep = None
    for k in range(40):
        p2, r2 = oracle.fk(s2)
        cap = (p2[col['body']] + r2[col['body']] @ col['verts'][0], p2[col['body']] + r2[col['body']] @ col['verts'][1], col['radius'])
        if k % 8 == 0:
            anchor, _ = oracle.ee_pose(s2)
        xr, vr, rep = numpy_substep(t, xr, vr, -9.81, dt, anchor, capsule=cap, friction=col['friction'])
        _advance_rigid_one_substep(blob, oracle, s2)
    assert len(rep) >= 5 and len(con) == len(rep)
    assert np.abs(cloth[0] - xr).max() < 5e-6 and np.abs(cloth[1] - vr).max() < 2e-3
    want = np.array([np.concatenate(rep[i]) for i in sorted(rep)])
    assert np.allclose(con[:, :3], want[:, :3], atol=5e-6) and np.allclose(con[:, 3:], want[:, 3:], rtol=2e-3, atol=3e-5)      # forces of 0.4 ... 3 mN on this 1.2 g patch: sums of nearly cancelling corrections


def _advance_rigid_one_substep(blob, oracle, s):
    """one INTERNAL substep of the rigid scene (no garment): a blob with SIM_SUBSTEPS = 1 and DT / 8 steps exactly one substep per settle call"""
    key = '_sub'
    if not hasattr(_advance_rigid_one_substep, key):
        from assistive_gym_amd.blob import ModelBlob
        from oracle_lib import Oracle
        w = blob.words.copy()
        w[L.H['SIM_SUBSTEPS']] = 1
        w.view(np.float32)[blob.h['OFF_PARAMS'] + L.P['DT']] = np.float32(0.02) / np.float32(8)
        setattr(_advance_rigid_one_substep, key, Oracle(ModelBlob(w, blob.meta)))
    getattr(_advance_rigid_one_substep, key).settle(s, 1)
