"""Mesh readers for the model compiler (host side, reset/compile time only).

The reference hands these files to PyBullet's importers:
  * multi-object Wavefront OBJ (``*_vhacd.obj``) -> one convex hull per ``o`` group
    (reference call sites: assistive_gym/envs/agents/tool.py:26-34,
    assistive_gym/envs/human_creation.py:103-105, assets/dinnerware/bowl.urdf:20-25)
  * Collada ``.dae`` collision meshes of the Jaco links -> a single convex hull of all
    vertices (assets/jaco/j2s7s300_gym.urdf:96-100 ...).

Collada ``up_axis`` is deliberately ignored: the URDF joint origins / inertial origins are
authored against the raw mesh coordinates (e.g. arm_half_1.dae spans y in [-0.205, 0] and
joint_3 sits at xyz="0 -0.205 0", j2s7s300_gym.urdf:196-200), which is also what the ROS
loaders do.  [BULLET-UNVERIFIED]: Bullet's URDF importer is believed to skip the up-axis
rotation as well.
"""
import re
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull


def load_obj_groups(path, scale=1.0):
    """Return a list of (n_i, 3) float64 vertex arrays, one per ``o`` group."""
    groups, cur = [], None
    with open(path) as fh:
        for line in fh:
            if line.startswith('o '):
                cur = []
                groups.append(cur)
            elif line.startswith('v '):
                if cur is None:
                    cur = []
                    groups.append(cur)
                p = line.split()
                cur.append((float(p[1]), float(p[2]), float(p[3])))
    scale = np.asarray(scale, dtype=np.float64)
    return [np.asarray(g, dtype=np.float64) * scale for g in groups if len(g) > 0]


def load_stl_vertices(path):
    """All vertices of a binary STL (the PR2 link meshes, assets/PR2/meshes/*/*.stl): PyBullet turns such a mesh into one
    convex hull of its vertices, like the Collada ones."""
    import struct
    d = open(path, 'rb').read()
    n = struct.unpack_from('<I', d, 80)[0]
    assert 84 + 50 * n == len(d), 'only binary STL files are supported: ' + path
    tri = np.frombuffer(d, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]), count=n, offset=84)
    return np.unique(tri['v'].reshape(-1, 3).astype(np.float64), axis=0)


def load_dae_vertices(path):
    """All position vertices of every geometry instanced by the visual scene, node matrices applied."""
    text = open(path).read()
    text = re.sub(r'xmlns="[^"]+"', '', text, count=1)
    root = ET.fromstring(text)
    geoms = {}
    for g in root.iter('geometry'):
        mesh = g.find('mesh')
        if mesh is None:
            continue
        verts_el = mesh.find('vertices')
        src_id = None
        if verts_el is not None:
            for inp in verts_el.findall('input'):
                if inp.get('semantic') == 'POSITION':
                    src_id = inp.get('source').lstrip('#')
        arr = None
        for src in mesh.findall('source'):
            if src_id is None or src.get('id') == src_id:
                fa = src.find('float_array')
                if fa is not None:
                    arr = np.array(fa.text.split(), dtype=np.float64).reshape(-1, 3)
                    break
        if arr is not None:
            geoms[g.get('id')] = arr
    unit = 1.0
    u = root.find('asset/unit')
    if u is not None and u.get('meter'):
        unit = float(u.get('meter'))
    out = []

    def walk(node, M):
        m = node.find('matrix')
        if m is not None:
            M = M @ np.array(m.text.split(), dtype=np.float64).reshape(4, 4)
        for ig in node.findall('instance_geometry'):
            v = geoms.get(ig.get('url').lstrip('#'))
            if v is not None:
                out.append((v @ M[:3, :3].T + M[:3, 3]) * unit)
        for ch in node.findall('node'):
            walk(ch, M)

    for vs in root.iter('visual_scene'):
        for node in vs.findall('node'):
            walk(node, np.eye(4))
    if not out:
        out = list(geoms.values())
    return np.concatenate(out, axis=0)


def convex_hull_vertices(points):
    """Vertices of the convex hull of ``points`` (float64, unique, original coordinates)."""
    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    if len(pts) <= 4:
        return pts
    try:
        hull = ConvexHull(pts)
        return pts[np.sort(hull.vertices)]
    except Exception:
        return pts


def reduce_hull(verts, max_verts, seed=0):
    """Inner approximation of a convex hull with at most ``max_verts`` vertices.

    Starts from the axis-extreme vertices, then greedily adds the vertex that is farthest outside
    the current approximation until the budget is used.  Deterministic.  The reduced hull is a subset of the original hull vertices, so it is
    contained in the original shape (maximum inward error reported by ``hull_error``).
    """
    verts = np.asarray(verts, dtype=np.float64)
    if len(verts) <= max_verts:
        return verts
    # the 6 axis-extreme vertices first so the AABB is preserved exactly
    axes = np.concatenate([np.eye(3), -np.eye(3)], axis=0)
    chosen = []
    for d in axes:
        k = int(np.argmax(verts @ d))
        if k not in chosen:
            chosen.append(k)
    # greedy refinement: add the vertex with the largest distance outside the current hull
    while len(chosen) < max_verts:
        try:
            h = ConvexHull(verts[chosen])
        except Exception:
            break
        d = verts @ h.equations[:, :3].T + h.equations[:, 3]
        worst = d.max(axis=1)
        k = int(np.argmax(worst))
        if worst[k] <= 1e-9 or k in chosen:
            break
        chosen.append(k)
    return verts[sorted(chosen)]


def hull_error(full, reduced):
    """Max distance by which a vertex of ``full`` lies outside the hull of ``reduced``."""
    if len(reduced) < 4:
        return float('nan')
    h = ConvexHull(reduced)
    d = full @ h.equations[:, :3].T + h.equations[:, 3]
    return float(max(0.0, d.max(axis=1).max()))
