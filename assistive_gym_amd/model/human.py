"""Capsule human: restatement of the model data in assistive_gym/envs/human_creation.py:58-316.

Only the numbers are taken from the reference (dimensions :82-122 male, :133-173 female; link
tables :188-278; mass fractions; joint axes / limits); the representation is a plain kinematic
tree in PyBullet's final depth-first link numbering (the legend at human_creation.py:5-46 and
agents/human.py:5-58), so ``human.head == 23`` etc. keep their meaning.

Used at compile time (collision shapes per link, in the link frame) and at reset time (forward
kinematics of the posed, static human -> per-env world transforms of its collision bodies).
"""
import numpy as np

from . import xform as X

D = np.deg2rad

# creation order blocks, exactly as human_creation.py builds them (1-based parents, 0 = base)
def _tables(gender, limit_scale, rs=1.0, hs=1.0, cloth=False):
    if gender == 'male':
        m = 78.4
        dims = dict(
            chest=(0.127 * rs, 0.056), shoulders=(0.106 * rs, 0.253 / 8, 0.253 / 2.5 - 0.253 / 16),
            neck=(0.06 * rs, 0.124 * hs, (0.2565 - 0.1415 - 0.025) * hs),
            upperarm=(0.043 * rs, 0.279 * hs), forearm=(0.033 * rs, 0.257 * hs), hand=0.043 * rs,
            waist=(0.1205 * rs, 0.049), hips=(0.1335 * rs, 0.094, -0.08125 * hs),
            thigh=(0.08 * rs, 0.424 * hs), shin=(0.05 * rs, 0.403 * hs),
            foot=(0.05 * rs, 0.215 * hs, (0, -0.1, -0.025 * rs)))
        chest_p = [0, 0, 1.2455 * hs]
        shoulders_p = [0, 0, 0.1415 / 2 * hs]
        neck_p = [0, 0, 0.1515 * hs]
        head_p = [0, 0, (0.399 - 0.1415 - 0.1205) * hs]
        right_upperarm_p = [-0.106 * rs - 0.073, 0, 0]
        left_upperarm_p = [0.106 * rs + 0.073, 0, 0]
        forearm_p = [0, 0, -0.279 * hs]
        hand_p = [0, 0, -(0.033 * rs + 0.257 * hs)]
        waist_p = [0, 0, -0.156 * hs]
        hips_p = [0, 0, -0.08125 * hs]
        right_thigh_p = [-0.08 * rs - 0.009, 0, -0.08125 * hs]
        left_thigh_p = [0.08 * rs + 0.009, 0, -0.08125 * hs]
        shin_p = [0, 0, -0.424 * hs]
        foot_p = [0, 0, -0.403 * hs - 0.025]
        head_mesh = ('head_female_male/BaseHeadMeshes_v5_male_cropped_reduced_compressed_vhacd.obj',
                     [0.09, 0.08, -0.07 + 0.01])
    else:
        m = 62.5
        dims = dict(
            chest=(0.127 * rs, 0.01), shoulders=(0.092 * rs, 0.225 / 8, 0.225 / 2.5 - 0.225 / 16),
            neck=(0.05 * rs, 0.121 * hs, (0.2565 - 0.1415 - 0.025) * hs),
            upperarm=(0.0355 * rs, 0.264 * hs), forearm=(0.027 * rs, 0.234 * hs), hand=0.0355 * rs,
            waist=(0.11 * rs, 0.009), hips=(0.127 * rs, 0.117, -0.15 / 2 * hs),
            thigh=(0.0775 * rs, 0.391 * hs), shin=(0.045 * rs, 0.367 * hs),
            foot=(0.045 * rs, 0.195 * hs, (0, -0.09, -0.0225 * rs)))
        chest_p = [0, 0, 1.148 * hs]
        shoulders_p = [0, 0, 0.132 / 2 * hs]
        neck_p = [0, 0, 0.132 * hs]
        head_p = [0, 0, 0.12 * hs]
        right_upperarm_p = [-0.092 * rs - 0.067, 0, 0]
        left_upperarm_p = [0.092 * rs + 0.067, 0, 0]
        forearm_p = [0, 0, -0.264 * hs]
        hand_p = [0, 0, -(0.027 * rs + 0.234 * hs)]
        waist_p = [0, 0, -0.15 * hs]
        hips_p = [0, 0, -0.15 / 2 * hs]
        right_thigh_p = [-0.0775 * rs - 0.0145, 0, -0.15 / 2 * hs]
        left_thigh_p = [0.0775 * rs + 0.0145, 0, -0.15 / 2 * hs]
        shin_p = [0, 0, -0.391 * hs]
        foot_p = [0, 0, -0.367 * hs - 0.045 / 2]
        head_mesh = ('head_female_male/BaseHeadMeshes_v5_female_cropped_reduced_compressed_vhacd.obj',
                     [-0.089, -0.09, -0.07])
    jp = [0, 0, 0]
    mass, shape, pos, parent, jtype, axis, lo, hi = [], [], [], [], [], [], [], []
    ls = limit_scale
    # shoulders, neck, head (human_creation.py:188-200)
    mass += [0, 0, 0.05, 0, 0, 0.05, 0.01, 0, 0, 0.07]
    shape += [None, None, 'right_shoulders', None, None, 'left_shoulders', 'neck', None, None, 'head']
    pos += [shoulders_p, shoulders_p, jp, shoulders_p, shoulders_p, jp, neck_p, head_p, jp, jp]
    parent += [0, 1, 2, 0, 4, 5, 0, 7, 8, 9]
    jtype += ['r'] * 10
    axis += [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    lo += list(np.array([D(-10), D(-10), D(-35), D(-10), D(-30), D(-35), D(-10), D(-50), D(-34), D(-70)]) * ls)
    hi += list(np.array([D(10), D(30), D(35), D(10), D(10), D(35), D(20), D(50), D(34), D(70)]) * ls)
    # right arm (:202-218)
    mass += [0, 0, 0.033, 0, 0.019, 0, 0.0065]
    shape += [None, 'shoulder_cloth' if cloth else None, 'upperarm', 'elbow_cloth' if cloth else None, 'forearm',
              'wrist_cloth' if cloth else None, 'hand']
    pos += [right_upperarm_p, jp, jp, forearm_p, jp, hand_p, jp]
    parent += [3, 11, 12, 13, 14, 15, 16]
    jtype += ['r'] * 7
    axis += [[0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]]
    lo += list(np.array([D(5), D(-188), D(-90), D(-128), D(-90), D(-81), D(-27)]) * ls)
    hi += list(np.array([D(198), D(61), D(90), D(0), D(90), D(90), D(47)]) * ls)
    # left arm (:220-236)
    mass += [0, 0, 0.033, 0, 0.019, 0, 0.0065]
    shape += [None, 'shoulder_cloth' if cloth else None, 'upperarm', 'elbow_cloth' if cloth else None, 'forearm',
              'wrist_cloth' if cloth else None, 'hand']
    pos += [left_upperarm_p, jp, jp, forearm_p, jp, hand_p, jp]
    parent += [6, 18, 19, 20, 21, 22, 23]
    jtype += ['r'] * 7
    axis += [[0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]]
    lo += list(np.array([D(-198), D(-188), D(-90), D(-128), D(-90), D(-81), D(-47)]) * ls)
    hi += list(np.array([D(-5), D(61), D(90), D(0), D(90), D(90), D(27)]) * ls)
    # waist and hips (:238-250)
    mass += [0, 0, 0.13, 0.14]
    shape += ['waist', None, None, 'hips']
    pos += [waist_p, hips_p, jp, jp]
    parent += [0, 25, 26, 27]
    jtype += ['f', 'r', 'r', 'r']
    axis += [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    lo += [0, D(-75), D(-30), D(-30)]
    hi += [0, D(30), D(30), D(30)]
    # right leg (:252-264)
    mass += [0, 0, 0.105, 0.0475, 0, 0, 0.014]
    shape += [None, None, 'thigh', 'shin', None, None, 'foot']
    pos += [right_thigh_p, jp, jp, shin_p, foot_p, jp, jp]
    parent += [28, 29, 30, 31, 32, 33, 34]
    jtype += ['r'] * 7
    axis += [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    lo += [D(-127), D(-40), D(-45), 0, D(-35), D(-23), D(-43)]
    hi += [D(30), D(45), D(40), D(130), D(38), D(24), D(35)]
    # left leg (:266-278)
    mass += [0, 0, 0.105, 0.0475, 0, 0, 0.014]
    shape += [None, None, 'thigh', 'shin', None, None, 'foot']
    pos += [left_thigh_p, jp, jp, shin_p, foot_p, jp, jp]
    parent += [28, 36, 37, 38, 39, 40, 41]
    jtype += ['r'] * 7
    axis += [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    lo += [D(-127), D(-45), D(-40), 0, D(-35), D(-24), D(-35)]
    hi += [D(30), D(40), D(45), D(130), D(38), D(23), D(43)]
    return dict(m=m, dims=dims, chest_p=chest_p, head_mesh=head_mesh, mass=np.array(mass) * m, shape=shape,
                pos=np.array(pos, dtype=np.float64), parent=parent, jtype=jtype, axis=np.array(axis, dtype=np.float64),
                lo=np.array(lo), hi=np.array(hi))


class HumanModel:
    """Kinematic tree in PyBullet link numbering (pre-order DFS of the creation-order tree)."""

    def __init__(self, gender='male', limit_scale=1.0, cloth=False):
        t = _tables(gender, limit_scale, cloth=cloth)
        self.gender = gender
        self.total_mass = t['m']
        self.dims = t['dims']
        self.chest_p = np.array(t['chest_p'])
        self.head_mesh = t['head_mesh']
        n = len(t['parent'])
        children = {i: [] for i in range(n + 1)}   # creation ids are 1-based, 0 = base
        for c, p in enumerate(t['parent']):
            children[p].append(c + 1)
        order = []

        def dfs(i):
            for c in children[i]:
                order.append(c)
                dfs(c)
        dfs(0)
        new_of = {0: -1}
        for k, c in enumerate(order):
            new_of[c] = k
        self.n = n
        self.parent = np.array([new_of[t['parent'][c - 1]] for c in order])
        self.offset = np.array([t['pos'][c - 1] for c in order])
        self.axis = np.array([t['axis'][c - 1] for c in order])
        self.jtype = [t['jtype'][c - 1] for c in order]
        self.lower = np.array([t['lo'][c - 1] for c in order])
        self.upper = np.array([t['hi'][c - 1] for c in order])
        self.mass = np.array([t['mass'][c - 1] for c in order])
        self.shape = [t['shape'][c - 1] for c in order]

    # --- collision shapes, in the owning link frame ------------------------------------------
    def colliders(self):
        """Returns a list of (link, kind, data): kind 'capsule' -> (p0, p1, r); 'sphere' -> (c, r);
        'head' -> (mesh file, frame pos, frame quat, scale).  link == -1 is the base (chest)."""
        d = self.dims
        ex, ey = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
        ez = np.array([0, 0, 1.0])

        def cap(r, length, centre, ax):
            c = np.asarray(centre, dtype=np.float64)
            return ('capsule', (c - ax * length / 2, c + ax * length / 2, r))
        table = {
            'right_shoulders': cap(d['shoulders'][0], d['shoulders'][1], [-d['shoulders'][2], 0, 0], ex),
            'left_shoulders': cap(d['shoulders'][0], d['shoulders'][1], [d['shoulders'][2], 0, 0], ex),
            'neck': cap(d['neck'][0], d['neck'][1], [0, 0, d['neck'][2]], ez),
            'upperarm': cap(d['upperarm'][0], d['upperarm'][1], [0, 0, -d['upperarm'][1] / 2], ez),
            'forearm': cap(d['forearm'][0], d['forearm'][1], [0, 0, -d['forearm'][1] / 2], ez),
            'hand': ('sphere', (np.array([0, 0, -d['hand']]), d['hand'])),
            'waist': cap(d['waist'][0], d['waist'][1], [0, 0, 0], ex),
            'hips': cap(d['hips'][0], d['hips'][1], [0, 0, d['hips'][2]], ex),
            'thigh': cap(d['thigh'][0], d['thigh'][1], [0, 0, -d['thigh'][1] / 2], ez),
            'shin': cap(d['shin'][0], d['shin'][1], [0, 0, -d['shin'][1] / 2], ez),
            'foot': cap(d['foot'][0], d['foot'][1], d['foot'][2], ey),
            'shoulder_cloth': ('sphere', (np.zeros(3), d['upperarm'][0])),
            'elbow_cloth': ('sphere', (np.zeros(3), d['upperarm'][0])),
            'wrist_cloth': ('sphere', (np.zeros(3), d['forearm'][0])),
            'head': ('head', (self.head_mesh[0], np.array(self.head_mesh[1]), X.quat_from_rpy([np.pi / 2, 0, 0]), 0.89)),
        }
        out = [(-1,) + cap(d['chest'][0], d['chest'][1], [0, 0, 0], ex)]
        for i, s in enumerate(self.shape):
            if s is not None:
                out.append((i,) + table[s])
        return out

    # --- kinematics ----------------------------------------------------------------------------
    def clamp(self, q):
        """Agent.enforce_joint_limits (agents/agent.py:240-250) on a joint-angle vector."""
        q = np.array(q, dtype=np.float64)
        for i in range(self.n):
            if self.jtype[i] == 'r':
                q[i] = min(max(q[i], self.lower[i]), self.upper[i])
            else:
                q[i] = 0.0
        return q

    def fk(self, base_pos, base_quat, q):
        """World (pos, quat) of every link frame; link frame = joint frame, rotations identity at q=0."""
        pos = np.zeros((self.n, 3))
        quat = np.zeros((self.n, 4))
        for i in range(self.n):
            pp, pq = (base_pos, base_quat) if self.parent[i] < 0 else (pos[self.parent[i]], quat[self.parent[i]])
            jq = X.quat_from_axis_angle(self.axis[i], q[i]) if self.jtype[i] == 'r' else np.array([0, 0, 0, 1.0])
            pos[i], quat[i] = X.compose(pp, pq, self.offset[i], jq)
        return pos, quat
