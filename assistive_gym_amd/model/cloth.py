"""Cloth section of the model blob (include/agx_blob.h, AGX_CL_*): the garment DressingEnv.reset loads
(assistive_gym/envs/dressing.py:149-157: p.loadCloth(hospitalgown_reduced.obj, scale 1.4, mass 0.16, position, orientation,
anchors, collisionMargin 0.04) + p.clothParams(...)).

The fork's cloth API is Bullet's btSoftBody behind two extra PyBullet calls [BULLET-UNVERIFIED]; what this module fixes:
  * node order = the OBJ's vertices in order of first appearance in its face list (tinyobj re-indexes (v, vn) pairs that way).
    Check: with that order the four anchor nodes 2086, 2087, 2088, 2041 lie within 2 cm of each other and 1.2 cm from
    `cloth_orig_pos` (dressing.py:148) under world = scale * (R v + position); in file order they are 75 cm apart;
  * links = the unique edges of the triangles, in order of first appearance (btSoftBodyHelpers::CreateFromTriMesh), no
    bending links; here they are additionally sorted into colour classes (links of a class share no node) so that a class
    can be relaxed in parallel -- a different Gauss-Seidel order than Bullet's plain list order;
  * node mass = total mass / node count (btSoftBody::setTotalMass(mass, fromfaces=false));
  * node area = mean rest area of the incident faces (btSoftBody::updateArea).
"""
import numpy as np

from . import xform as X
from .compiler import CL, CP, CLOTH_MAX_COLORS, CLOTH_THREADS


def load_obj_first_appearance(path):
    """vertices re-indexed in order of first appearance in the face list; faces in file order"""
    V, F = [], []
    for line in open(path):
        if line.startswith('v '):
            V.append([float(t) for t in line.split()[1:4]])
        elif line.startswith('f '):
            F.append([int(t.split('/')[0]) - 1 for t in line.split()[1:4]])
    V, F = np.array(V), np.array(F)
    new_of, order = {}, []
    for f in F:
        for a in f:
            if a not in new_of:
                new_of[a] = len(order)
                order.append(a)
    return V[order], np.array([[new_of[a] for a in f] for f in F])


def mesh_links(faces):
    """unique edges in order of first appearance: (0,1), (1,2), (2,0) of every face (CreateFromTriMesh)"""
    seen, links = set(), []
    for f in faces:
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            key = (min(a, b), max(a, b))
            if key not in seen:
                seen.add(key)
                links.append((int(a), int(b)))
    return links


def colour_links(links, n_nodes, cap):
    """greedy colouring in list order: the smallest class that holds neither node and has room"""
    used = []                      # per class: set of nodes
    cls = []
    for a, b in links:
        for c, u in enumerate(used):
            if a not in u and b not in u and len(u) < 2 * cap:
                break
        else:
            used.append(set())
            c = len(used) - 1
        used[c].update((a, b))
        cls.append(c)
    return np.array(cls), len(used)


def colour_links_balanced(links, cap):
    """the links of one patch in as few classes of at most `cap` links as a greedy search finds: for K = ceil(n / cap), K + 1, ... every
    link (longest-waiting first: by the degree of its nodes) goes to the feasible class with the fewest links; the first K that places
    all of them wins.  Returns (class of every link, K)."""
    n = len(links)
    if n == 0:
        return np.zeros(0, dtype=int), 0
    deg = {}
    for a, b in links:
        deg[a] = deg.get(a, 0) + 1; deg[b] = deg.get(b, 0) + 1
    order = sorted(range(n), key=lambda k: -(deg[links[k][0]] + deg[links[k][1]]))
    for K in range((n + cap - 1) // cap, n + 1):
        used, cnt, cls, ok = [set() for _ in range(K)], [0] * K, [0] * n, True
        for k in order:
            a, b = links[k]
            best = None
            for c in range(K):
                if cnt[c] < cap and a not in used[c] and b not in used[c] and (best is None or cnt[c] < cnt[best]):
                    best = c
            if best is None:
                ok = False
                break
            used[best].update((a, b)); cnt[best] += 1; cls[k] = best
        if ok:
            return np.array(cls), K
    raise AssertionError('unreachable')


def bank_schedule(class_links, lanes=32, seeds=3):
    """Order of the links of ONE colour class for the cloth kernel (thread t relaxes slot t): x lives in LDS as float[NN][3], so the
    lanes of a 32-lane bank group read / write without conflict exactly when their node indices are distinct mod 32 (bank =
    (3 node + k) mod 32, 3 is a unit mod 32); every extra node on a busy bank costs one more LDS cycle for the group.  Greedy: the
    links are dealt, in random order, to the group (and the endpoint order: the relaxation is symmetric) where they raise the group's
    worst bank multiplicity least; best of a few seeds.  The links of a class share no node, so their order changes no result.
    Measured before: 54 % of the cloth kernel's LDS cycles were bank conflicts (profiles/r03_traffic_dressing.json,
    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE); sorted by first node the link phases paid 2.1 extra cycles per group access, this
    schedule 0.7.  Returns (ordered links, extra cycles)."""
    n = len(class_links)
    G = (n + lanes - 1) // lanes
    best = None
    for seed in range(seeds):
        rng = np.random.RandomState(seed)
        cap = [lanes] * G
        if G:
            cap[-1] = n - lanes * (G - 1)
        cnt_a, cnt_b, mem = np.zeros((G, lanes), int), np.zeros((G, lanes), int), [[] for _ in range(G)]
        for idx in rng.permutation(n):
            a, b = class_links[idx]
            pick = None
            for g in range(G):
                if len(mem[g]) >= cap[g]:
                    continue
                for x, y in ((a, b), (b, a)):
                    ca = max(cnt_a[g][x % lanes] + 1 - max(cnt_a[g].max(), 1), 0)
                    cb = max(cnt_b[g][y % lanes] + 1 - max(cnt_b[g].max(), 1), 0)
                    key = (ca + cb, cnt_a[g][x % lanes] + cnt_b[g][y % lanes], len(mem[g]))
                    if pick is None or key < pick[0]:
                        pick = (key, g, (x, y))
            _, g, (x, y) = pick
            mem[g].append((int(x), int(y))); cnt_a[g][x % lanes] += 1; cnt_b[g][y % lanes] += 1
        extra = sum(max(cnt_a[g].max() - 1, 0) + max(cnt_b[g].max() - 1, 0) for g in range(G))
        if best is None or extra < best[1]:
            best = ([l for g in mem for l in g], extra)
    return best if best else ([], 0)


def hull_planes(verts):
    """outward unit normals and offsets (n.x = off on the face) of the convex hull of the vertices, coplanar facets merged"""
    from scipy.spatial import ConvexHull
    v = np.asarray(verts, dtype=np.float64)
    try:
        eq = ConvexHull(v).equations
    except Exception:
        lo, hi = v.min(0), v.max(0)            # a flat or degenerate set: its bounding box
        return np.array([[1, 0, 0, hi[0]], [-1, 0, 0, -lo[0]], [0, 1, 0, hi[1]], [0, -1, 0, -lo[1]], [0, 0, 1, hi[2]], [0, 0, -1, -lo[2]]], dtype=np.float64)
    out, keys = [], set()
    for n0, n1, n2, d in eq:
        key = (round(n0, 5), round(n1, 5), round(n2, 5), round(-d, 6))
        if key not in keys:
            keys.add(key)
            out.append([n0, n1, n2, -d])
    return np.array(out)


def compile_cloth(obj_path, scale, position, rpy, anchors, anchor_body_pos, tri1, tri2, params, colliders, shape_ids, gender_of=lambda ci: 0):
    """uint32 words of the cloth section.  colliders: Scene.colliders; shape_ids: the colliders the cloth is tested against;
    gender_of(collider) = 0 always present, 1 male human only, 2 female human only."""
    v, faces = load_obj_first_appearance(obj_path)
    R = X.quat_to_mat(X.quat_from_rpy(rpy))
    x0 = scale * (v @ R.T + np.asarray(position))                    # see module docstring
    nn = len(x0)
    links = [(min(a, b), max(a, b)) for a, b in mesh_links(faces)]
    # ownership order of the cloth kernel: Morton order of the rest positions (10 bits per axis); wave w owns 256 consecutive nodes of it
    g = np.floor((x0 - x0.min(0)) / (np.ptp(x0, axis=0).max() + 1e-9) * 1023).astype(np.int64)

    def spread(v):
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    morton = spread(g[:, 0]) | (spread(g[:, 1]) << 1) | (spread(g[:, 2]) << 2)
    perm = np.full(4096, -1, dtype=np.int64)
    perm[:nn] = np.argsort(morton, kind='stable')
    assert nn <= 4096
    # Link schedule.  A link whose two nodes belong to the same patch (the 256 nodes of a wave) is relaxed by that wave, colour by colour,
    # with no workgroup barrier: patches share no node.  Only the links BETWEEN patches (a ninth of them) keep workgroup-wide classes.
    # Within a class the order is free (no two links share a node): a bank-conflict-aware order for the cloth kernel (bank_schedule).
    W = CLOTH_THREADS // 64
    patch_of = np.full(nn, -1)
    for w in range(W):
        own = perm[w * (4096 // W):(w + 1) * (4096 // W)]
        patch_of[own[own >= 0]] = w
    inner = [[l for l in links if patch_of[l[0]] == w and patch_of[l[1]] == w] for w in range(W)]
    cross = [l for l in links if patch_of[l[0]] != patch_of[l[1]]]
    inner_cls = [colour_links_balanced(li, 64) for li in inner]
    npatch_color = max(k for _, k in inner_cls)
    sched, color_off, bank_extra = [], [0], 0
    for w in range(W):
        cls_w, _ = inner_cls[w]
        for c in range(npatch_color):
            order_c, extra_c = bank_schedule(sorted(l for l, k in zip(inner[w], cls_w) if k == c))
            assert len(order_c) <= 64
            sched += order_c + [None] * (64 - len(order_c)); bank_extra += extra_c
            color_off.append(len(sched))
    cls, ncross = colour_links(cross, nn, 1024)                  # the cloth kernel relaxes one link of a class per thread
    assert ncross <= CLOTH_MAX_COLORS, ncross
    for c in range(ncross):
        order_c, extra_c = bank_schedule(sorted(l for l, k in zip(cross, cls) if k == c))
        sched += order_c; bank_extra += extra_c
        color_off.append(len(sched))
    ncolor = W * npatch_color + ncross
    links = sched
    n_real = sum(1 for l in links if l is not None)
    assert n_real == len(inner[0]) + sum(len(t) for t in inner[1:]) + len(cross)
    max_per = max([color_off[c + 1] - color_off[c] for c in range(W * npatch_color, ncolor)] + [0])
    assert max_per <= 1024
    rest2 = np.array([np.sum((x0[l[0]] - x0[l[1]]) ** 2) if l is not None else 0.0 for l in links])
    # incident faces per node, in face order, rotated so that the node comes first (same cross product)
    inc = [[] for _ in range(nn)]
    area_sum, cnt = np.zeros(nn), np.zeros(nn)
    for f in faces:
        a, b, c = (int(t) for t in f)
        ra = 0.5 * np.linalg.norm(np.cross(x0[b] - x0[a], x0[c] - x0[a]))
        for i, j, k in ((a, b, c), (b, c, a), (c, a, b)):
            inc[i].append(j | (k << 16))
            area_sum[i] += ra
            cnt[i] += 1
    area = np.where(cnt > 0, area_sum / np.maximum(cnt, 1), 0.0)
    node_first = np.concatenate([[0], np.cumsum([len(t) for t in inc])]).astype(np.int64)
    face_entries = np.array([e for t in inc for e in t], dtype=np.int64)
    # rigid shapes: capsule / sphere cores are evaluated exactly, hulls through their face planes
    planes, shapes = [], []
    for ci in shape_ids:
        c = colliders[ci]
        if len(c['verts']) <= 2:
            shapes.append([ci, 0, 0, gender_of(ci)])
        else:
            pl = hull_planes(c['verts']).tolist()
            pl += [pl[-1]] * ((-len(pl)) % 4)      # the cloth kernel evaluates four planes per scalar load: the last plane repeated (ties keep the first)
            shapes.append([ci, len(planes), len(pl), gender_of(ci)])
            planes.extend(pl)
    planes = np.array(planes, dtype=np.float64).reshape(-1, 4)
    off, cur = {}, CL['HDR']
    cur += (-cur) % 4      # (the PLANE array is read as 16-byte words: the section itself starts on a 16-byte boundary, see pack())
    for name, size in (('COLOR', ncolor + 1), ('LINK', 2 * len(links)), ('NODE', 2 * (nn + 1)), ('FACE', len(face_entries)), ('X0', 3 * nn),
                       ('ANCHOR', 4 * len(anchors)), ('SHAPE', 4 * len(shapes)), ('PLANE', 4 * len(planes)), ('PARAM', CP['COUNT']), ('PERM', 4096)):
        if name == 'PLANE':
            cur += (-cur) % 4
        off[name] = cur
        cur += size
    f = np.zeros(cur, dtype=np.float32)
    i = f.view(np.int32)
    i[CL['NN']], i[CL['NL']], i[CL['NCOLOR']], i[CL['NANCHOR']], i[CL['NSHAPE']] = nn, len(links), ncolor, len(anchors), len(shapes)
    for name in ('COLOR', 'LINK', 'NODE', 'FACE', 'X0', 'ANCHOR', 'SHAPE', 'PLANE', 'PARAM', 'PERM'):
        i[CL['OFF_' + name]] = off[name]
    i[CL['TRI']:CL['TRI'] + 6] = list(tri1) + list(tri2)
    i[CL['MAX_LINKS_PER_COLOR']] = max_per
    i[CL['NPATCH_COLOR']] = npatch_color
    i[off['PERM']:off['PERM'] + 4096] = perm
    i[off['COLOR']:off['COLOR'] + ncolor + 1] = color_off
    for k, l in enumerate(links):
        i[off['LINK'] + 2 * k] = (l[0] | (l[1] << 16)) if l is not None else -1          # -1: an empty slot of the bank schedule
        f[off['LINK'] + 2 * k + 1] = rest2[k]
    for n in range(nn + 1):
        i[off['NODE'] + 2 * n] = node_first[n]
        f[off['NODE'] + 2 * n + 1] = area[n] if n < nn else 0.0
    i[off['FACE']:off['FACE'] + len(face_entries)] = face_entries
    f[off['X0']:off['X0'] + 3 * nn] = x0.astype(np.float32).ravel()
    for k, n in enumerate(anchors):
        i[off['ANCHOR'] + 4 * k] = n
        f[off['ANCHOR'] + 4 * k + 1:off['ANCHOR'] + 4 * k + 4] = x0[n] - np.asarray(anchor_body_pos)
    i[off['SHAPE']:off['SHAPE'] + 4 * len(shapes)] = np.array(shapes, dtype=np.int64).ravel() if shapes else []
    f[off['PLANE']:off['PLANE'] + 4 * len(planes)] = planes.astype(np.float32).ravel()
    pv = f[off['PARAM']:off['PARAM'] + CP['COUNT']]
    for k, val in params.items():
        if k != 'MASS':
            pv[CP[k]] = val
    pv[CP['NODE_IM']] = nn / params['MASS']                          # setTotalMass(mass, fromfaces=false): equal node masses
    meta = dict(nodes=nn, links=n_real, link_slots=len(links), link_bank_extra_cycles=int(bank_extra), colors=ncolor, patch_colors=npatch_color, cross_colors=ncross, cross_links=len(cross), faces=len(faces), shapes=len(shapes), planes=len(planes), max_links_per_color=max_per)
    return f.view(np.uint32).copy(), meta


def compile_particles(x0, radius, total_mass, params, colliders, shape_ids, gender_of=lambda ci: 0):
    """uint32 words of a particle section in the garment's format (AGX_CL_PARTICLES = 1): the water of the drinking task (drinking.py:160-170) --
    nodes without links, faces or anchors, spheres of `radius` (the section's MARGIN) tested against the same kind of shape table.
    x0: rest offsets of the particles (the 4 x 4 x 4 grid relative to the cup, drinking.py:163-167)."""
    x0 = np.asarray(x0, dtype=np.float64)
    nn = len(x0)
    assert nn <= 64
    planes, shapes = [], []
    for ci in shape_ids:
        c = colliders[ci]
        if len(c['verts']) <= 2:
            shapes.append([ci, 0, 0, gender_of(ci)])
        else:
            pl = hull_planes(c['verts']).tolist()
            pl += [pl[-1]] * ((-len(pl)) % 4)
            shapes.append([ci, len(planes), len(pl), gender_of(ci)])
            planes.extend(pl)
    planes = np.array(planes, dtype=np.float64).reshape(-1, 4)
    off, cur = {}, CL['HDR']
    cur += (-cur) % 4
    for name, size in (('COLOR', 1), ('LINK', 0), ('NODE', 2 * (nn + 1)), ('FACE', 0), ('X0', 3 * nn), ('ANCHOR', 0), ('SHAPE', 4 * len(shapes)),
                       ('PLANE', 4 * len(planes)), ('PARAM', CP['COUNT']), ('PERM', 4096)):
        if name == 'PLANE':
            cur += (-cur) % 4
        off[name] = cur
        cur += size
    f = np.zeros(cur, dtype=np.float32)
    i = f.view(np.int32)
    i[CL['NN']], i[CL['NL']], i[CL['NCOLOR']], i[CL['NANCHOR']], i[CL['NSHAPE']], i[CL['PARTICLES']] = nn, 0, 0, 0, len(shapes), 1
    for name in ('COLOR', 'LINK', 'NODE', 'FACE', 'X0', 'ANCHOR', 'SHAPE', 'PLANE', 'PARAM', 'PERM'):
        i[CL['OFF_' + name]] = off[name]
    perm = np.full(4096, -1, dtype=np.int64)
    perm[:nn] = np.arange(nn)
    i[off['PERM']:off['PERM'] + 4096] = perm
    f[off['X0']:off['X0'] + 3 * nn] = x0.astype(np.float32).ravel()
    i[off['SHAPE']:off['SHAPE'] + 4 * len(shapes)] = np.array(shapes, dtype=np.int64).ravel() if shapes else []
    f[off['PLANE']:off['PLANE'] + 4 * len(planes)] = planes.astype(np.float32).ravel()
    pv = f[off['PARAM']:off['PARAM'] + CP['COUNT']]
    for k, val in params.items():
        pv[CP[k]] = val
    pv[CP['MARGIN']] = radius
    pv[CP['NODE_IM']] = nn / total_mass
    return f.view(np.uint32).copy(), dict(nodes=nn, links=0, shapes=len(shapes), planes=len(planes), particles=True)
