"""Reference bridge (TEST INFRASTRUCTURE): the reference's OWN Python half executed on top of the CPU oracle's physics.

The reference (`/root/reference/assistive_gym`, pure Python) talks to its physics engine through ~25 `pybullet` functions.
PyBullet is not installable here, but everything ABOVE that API is runnable: this package provides a module that is installed
as `pybullet` (plus inert `gym` / `ray` / `keras` / `screeninfo` stubs), backs the calls the step path makes
(`getJointStates`, `getLinkState`, `getContactPoints`, `getClosestPoints`, `setJointMotorControlArray`, `resetJointState`,
`stepSimulation`, ...) by one persistent oracle world (`agxo_world_*`, oracle/agx_oracle.c), imports the reference's env classes
unmodified, and lets THEIR `step()` -- `AssistiveEnv.take_step` (env.py:174-235), `Agent.enforce_joint_limits`
(agent.py:240-250), `Human.enforce_realistic_joint_limits` (human.py:134-152), `<Task>Env._get_obs / get_total_force /
get_food_rewards / update_targets`, `human_preferences` (env.py:237-274), `Util.sleeve_on_arm_reward` (util.py:134-202) -- run.

What this pins: the whole Python half of the hot path against the reference's code executed
here; what it cannot pin: what happens INSIDE p.stepSimulation() (row a4, Bullet) -- that stays the oracle's restatement.

`reset()` is not executed (it builds the world through URDF / mesh loaders of the engine): `adopt()` wires the env object up
from a blob + state record the way the task's reset() leaves it, citing the lines it stands for.  The gains / forces the
reference's `Agent.control` passes to the engine are CHECKED against the blob (`World.gain_mismatches`).

Nothing here is product code; only tests/ and tests/diag/ import it, and it needs /root/reference (absent on the GPU box:
the fixtures it generates are committed under tests/golden/ref_*.npz together with tests/diag/make_reference_fixtures.py).
"""
import configparser
import ctypes as C
import importlib
import os
import sys
import types

import numpy as np

REF_ROOT = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# body unique ids of the facade
PLANE, ROBOT, HUMAN, FURNITURE, TOOL, TOOL2, TABLE, BOWL, ATTACH, CLOTH = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
FOOD0, WATER0, MARKER0 = 100, 200, 1000       # food / water particle k is body FOOD0 + k / WATER0 + k
TAG_ROBOT, TAG_TOOL, TAG_HUMAN, TAG_FOOD, TAG_BOWL, TAG_TABLE, TAG_PLANE, TAG_WHEELCHAIR, TAG_BED = 1, 2, 3, 4, 5, 6, 7, 8, 9
BODY_FREE0, BODY_HUMAN0 = 200, 300


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'assistive_gym', 'envs'))


# ------------------------------------------------------------------------------------------------ transforms (Bullet conventions)
def q_mul(a, b):
    x1, y1, z1, w1 = a; x2, y2, z2, w2 = b
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])


def q_rot(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ np.asarray(v, dtype=np.float64)


def q_mat(q):
    x, y, z, w = [float(t) for t in q]
    n = np.sqrt(x * x + y * y + z * z + w * w); x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def mat_q(m):
    """b3Matrix3x3::getRotation: what b3MultiplyTransforms / b3InvertTransform hand back (they work on 3x3 bases, so the sign
    of the returned quaternion is this function's, not that of a quaternion product)"""
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    t = [0.0] * 4
    if tr > 0:
        s = np.sqrt(tr + 1.0); t[3] = s * 0.5; s = 0.5 / s
        t[0] = (m[2, 1] - m[1, 2]) * s; t[1] = (m[0, 2] - m[2, 0]) * s; t[2] = (m[1, 0] - m[0, 1]) * s
    else:
        i = (2 if m[1, 1] < m[2, 2] else 1) if m[0, 0] < m[1, 1] else (2 if m[0, 0] < m[2, 2] else 0)
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0); t[i] = s * 0.5; s = 0.5 / s
        t[3] = (m[k, j] - m[j, k]) * s; t[j] = (m[j, i] + m[i, j]) * s; t[k] = (m[k, i] + m[i, k]) * s
    return np.array(t)


def q_from_euler(e):
    """btQuaternion::setEulerZYX(yaw = e[2], pitch = e[1], roll = e[0])"""
    r, p_, y = [0.5 * float(t) for t in e]
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p_), np.sin(p_), np.cos(y), np.sin(y)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def euler_from_q(q):
    x, y, z, w = [float(t) for t in q]
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * np.pi, 2 * np.arctan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * np.pi, 2 * np.arctan2(-x, y))
    return (np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), np.arcsin(sarg), np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z))


# ------------------------------------------------------------------------------------------------ the oracle world
def _lib():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_lib import lib
    L = lib()
    if not getattr(L, '_world_ready', False):
        L.agxo_world_create.restype = C.c_void_p
        L.agxo_world_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for n in ('agxo_world_store', 'agxo_world_free', 'agxo_world_joints', 'agxo_world_reset_joint', 'agxo_world_set_target',
                  'agxo_world_set_free_base', 'agxo_world_set_anchor', 'agxo_world_set_cloth_gravity', 'agxo_world_step', 'agxo_sleeve_reward'):
            getattr(L, n).restype = None
        L.agxo_world_reset_joint.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.agxo_world_set_target.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.agxo_world_set_cloth_gravity.argtypes = [C.c_void_p, C.c_double]
        for n in ('agxo_world_frame', 'agxo_world_contacts', 'agxo_world_closest', 'agxo_world_cloth', 'agxo_world_particle', 'agxo_world_particle_query'):
            getattr(L, n).restype = C.c_int
        L.agxo_world_set_particle.restype = None
        L.agxo_world_particle.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.agxo_world_set_particle.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.agxo_world_particle_query.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L._world_ready = True
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class World:
    """One environment of one model blob as the facade sees it: a persistent f64 oracle simulation + the id maps."""

    def __init__(self, blob, state, cloth=None):
        from oracle_lib import Oracle
        self.blob, self.L = blob, _lib()
        self.oracle = Oracle(blob)
        self.state0 = np.ascontiguousarray(state, dtype=np.float32).copy()
        self.cloth0 = None if cloth is None else np.ascontiguousarray(cloth, dtype=np.float32).copy()
        self.h = C.c_void_p(self.L.agxo_world_create(C.c_void_p(self.oracle.h), _p(self.state0), _p(self.cloth0)))
        v = blob.view(self.state0.reshape(1, -1))
        self.gender = int(v['gender'][0])
        self.plane_friction = float(v['plane_friction'][0])
        self.limit_scale = float(v['limit_scale'][0]) or 1.0
        self.human_kp, self.human_maxf = float(v['human_kp'][0]), float(v['human_maxf'][0])
        self.frozen = int(v['frozen'][0])
        g = self.gender
        # joints: (body, PyBullet joint index) -> DoF
        self.dof_of = {}
        for d in range(blob.ndof):
            self.dof_of[(ROBOT if d < blob.nrobot else HUMAN, blob.robot_i(d, 'PB_INDEX', g))] = d
        self.human_body_of_link = {int(l): i for i, l in enumerate(blob.meta.get('human_bodies', []))}
        self.tool_body = blob.h['TOOL_BODY']
        self.tool2_body = blob.task_i('TOOL2_BODY')
        self.food0, self.nfood = blob.h['FOOD0'], blob.nfood
        oc = blob.h['OFF_CLOTH']
        self.nwater = int(blob.i[oc + 0]) if (oc and blob.task_kind == 5) else 0      # AGX_CL_NN of a particle section (drinking)
        self.bowl_body = None
        for b in range(blob.nfree):
            if int(blob.i[blob.h['OFF_FREE'] + b * 16 + 12]) == 2:
                self.bowl_body = b
        # colliders -> (body id, link)
        self.coll = []
        rng = blob.meta.get('ranges', {})
        other = rng.get('human_female' if g == 0 else 'human_male', [0, 0])
        for c in range(blob.h['NCOLL']):
            k = blob.collider(c)
            tag, body = k['tag'], k['body']
            if tag == TAG_ROBOT: bid = ROBOT
            elif tag == TAG_TOOL: bid = TOOL2 if (self.tool2_body > 0 and body == BODY_FREE0 + self.tool2_body) else TOOL
            elif tag == TAG_HUMAN: bid = HUMAN if not (other[0] <= c < other[1]) else None       # the other gender's shapes are not in the world
            elif tag == TAG_FOOD: bid = FOOD0 + (body - BODY_FREE0 - self.food0)
            elif tag == TAG_BOWL: bid = BOWL
            elif tag == TAG_TABLE: bid = TABLE
            elif tag == TAG_PLANE: bid = PLANE
            else: bid = FURNITURE
            self.coll.append((bid, k['link']))
        self.ee_links = {}          # robot link -> tool index, set by adopt()
        self.markers, self.next_marker = {}, MARKER0     # visual-only bodies (targets): body id -> position
        self.gain_mismatches = []   # (body, joint, what, passed, blob)
        self.ignored = []           # calls that address something the model does not simulate
        self.steps = 0
        self.adopting, self.adopt_resets = False, []

    def close(self):
        if self.h:
            self.L.agxo_world_free(self.h); self.h = None

    # ---- state
    def joints(self):
        n = self.blob.ndof
        q, qd, qt = np.zeros(n), np.zeros(n), np.zeros(n)
        self.L.agxo_world_joints(self.h, _p(q), _p(qd), _p(qt))
        return q, qd, qt

    def store(self):
        st = self.state0.copy()
        cl = None if self.cloth0 is None else self.cloth0.copy()
        self.L.agxo_world_store(self.h, _p(st), _p(cl))
        return st, cl

    def frame(self, kind, index, vel=False):
        pos, quat, lin, ang = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3)
        ok = self.L.agxo_world_frame(self.h, C.c_int(kind), C.c_int(index), _p(pos), _p(quat), _p(lin), _p(ang))
        assert ok, ('no such frame', kind, index)
        return (pos, quat, lin, ang) if vel else (pos, quat)

    def link_frame(self, body, link, vel=False):
        if body == ROBOT:
            if link == -1: return self.frame(3, 0, vel)
            if link in self.ee_links: return self.frame(4, self.ee_links[link], vel)
            if (ROBOT, link) in self.dof_of: return self.frame(0, self.dof_of[(ROBOT, link)], vel)
        elif body == HUMAN:
            if (HUMAN, link) in self.dof_of: return self.frame(0, self.dof_of[(HUMAN, link)], vel)
            if link in self.human_body_of_link: return self.frame(2, self.human_body_of_link[link], vel)
        elif body in (TOOL, TOOL2):
            tb = self.tool_body if body == TOOL else self.tool2_body
            if link == -1: return self.frame(1, tb, vel)
            if link == 1: return self.frame(5, tb, vel)          # AGX_T_TOOL_OBS_*: link 1 of the wiper / the scratcher
        elif body == BOWL and link == -1:
            return self.frame(1, self.bowl_body, vel)
        elif FOOD0 <= body < FOOD0 + self.nfood and link == -1:
            return self.frame(1, self.food0 + body - FOOD0, vel)
        elif WATER0 <= body < WATER0 + self.nwater and link == -1:
            pos, lin = np.zeros(3), np.zeros(3)
            assert self.L.agxo_world_particle(self.h, body - WATER0, _p(pos), _p(lin))
            return (pos, np.array([0, 0, 0, 1.0]), lin, np.zeros(3)) if vel else (pos, np.array([0, 0, 0, 1.0]))
        elif body in self.markers and link == -1:
            z = np.zeros(3)
            return (self.markers[body], np.array([0, 0, 0, 1.0]), z, z) if vel else (self.markers[body], np.array([0, 0, 0, 1.0]))
        raise KeyError('the model has no frame for body %d link %d' % (body, link))

    def contacts(self):
        out = np.zeros((512, 16))
        n = self.L.agxo_world_contacts(self.h, _p(out), C.c_int(512))
        return out[:n]

    def closest(self, ca, cb, dist):
        ca, cb = np.ascontiguousarray(ca, dtype=np.int32), np.ascontiguousarray(cb, dtype=np.int32)
        out = np.zeros((max(1, len(ca) * len(cb)), 9))
        n = self.L.agxo_world_closest(self.h, _p(ca), C.c_int(len(ca)), _p(cb), C.c_int(len(cb)), C.c_double(dist), _p(out), C.c_int(len(out)))
        return out[:n]

    def colliders_of(self, body, link=None):
        return [c for c, (b, l) in enumerate(self.coll) if b == body and (link is None or l == link)]

    def dof_limits(self, d):
        b, g = self.blob, self.gender
        if not b.robot_i(d, 'HAS_LIMIT', g):
            return 0.0, -1.0
        sc = self.limit_scale if (b.robot_i(d, 'KIND', g) & 3) == 1 else 1.0
        return b.robot_f(d, 'LOWER', gender=g) * sc, b.robot_f(d, 'UPPER', gender=g) * sc


# ------------------------------------------------------------------------------------------------ the `pybullet` module
def make_pybullet(get_world):
    """A module object with the pybullet names the reference's step path uses; `get_world()` returns the World in use."""
    p = types.ModuleType('pybullet')
    p.DIRECT, p.GUI = 2, 1
    p.POSITION_CONTROL, p.VELOCITY_CONTROL, p.TORQUE_CONTROL = 2, 0, 1
    p.JOINT_REVOLUTE, p.JOINT_PRISMATIC, p.JOINT_FIXED = 0, 1, 4
    p.GEOM_SPHERE, p.GEOM_BOX, p.GEOM_CYLINDER, p.GEOM_MESH, p.GEOM_CAPSULE = 2, 3, 4, 5, 7
    p.URDF_USE_SELF_COLLISION, p.URDF_USE_INERTIA_FROM_FILE = 8, 2
    p.COV_ENABLE_RENDERING, p.COV_ENABLE_GUI, p.COV_ENABLE_MOUSE_PICKING = 7, 1, 9
    W = get_world

    def _noop(*a, **k): return None
    for n in ('resetSimulation', 'disconnect', 'resetDebugVisualizerCamera', 'configureDebugVisualizer', 'setTimeStep', 'setRealTimeSimulation',
              'setGravity', 'changeVisualShape', 'changeDynamics', 'setPhysicsEngineParameter', 'changeConstraint',
              'enableJointForceTorqueSensor', 'setJointMotorControl2'):
        setattr(p, n, _noop)
    p.connect = lambda *a, **k: 0
    def _new_marker(pos):
        w = W(); w.next_marker += 1
        w.markers[w.next_marker] = np.asarray(pos, dtype=np.float64).copy()
        return w.next_marker
    # world-building calls are recorded (RECORD) so that HumanCreation.create_human (human_creation.py:58-316) can be executed and its
    # arguments compared with model/human.py; they build nothing
    def createCollisionShape(shapeType=None, *a, **k):
        RECORD['shapes'].append(dict(k, shapeType=shapeType if shapeType is not None else k.get('shapeType')))
        return len(RECORD['shapes']) - 1
    p.createCollisionShape = createCollisionShape
    p.createVisualShape = lambda *a, **k: -1

    def setCollisionFilterPair(a, b, la, lb, enable, physicsClientId=0):
        RECORD['filter'].append((a, b, la, lb, int(enable)))
    p.setCollisionFilterPair = setCollisionFilterPair

    def createMultiBody(baseMass=0, baseCollisionShapeIndex=-1, baseVisualShapeIndex=-1, basePosition=(0, 0, 0), batchPositions=None, **k):
        if 'linkMasses' in k:
            RECORD['bodies'].append(dict(k, baseMass=baseMass, baseCollisionShapeIndex=baseCollisionShapeIndex, basePosition=basePosition))
            return HUMAN
        if batchPositions is not None:
            last = None
            for bp in batchPositions:
                last = _new_marker(np.asarray(basePosition, dtype=np.float64) + np.asarray(bp, dtype=np.float64))
            return last
        return _new_marker(basePosition)
    p.createMultiBody = createMultiBody

    p.getQuaternionFromEuler = lambda e, physicsClientId=0: tuple(q_from_euler(e))
    p.getEulerFromQuaternion = lambda q, physicsClientId=0: tuple(euler_from_q(q))

    def multiplyTransforms(positionA, orientationA, positionB, orientationB, physicsClientId=0):
        Ra, Rb = q_mat(orientationA), q_mat(orientationB)
        return tuple(np.asarray(positionA, dtype=np.float64) + Ra @ np.asarray(positionB, dtype=np.float64)), tuple(mat_q(Ra @ Rb))
    p.multiplyTransforms = multiplyTransforms

    def invertTransform(position, orientation, physicsClientId=0):
        Ri = q_mat(orientation).T
        return tuple(-(Ri @ np.asarray(position, dtype=np.float64))), tuple(mat_q(Ri))
    p.invertTransform = invertTransform

    def getNumJoints(body, physicsClientId=0):
        w = W()
        if w is None: return len(RECORD['bodies'][-1]['linkMasses']) if body == HUMAN and RECORD['bodies'] else 0
        if body == ROBOT: return w.n_robot_joints
        if body == HUMAN: return 42                      # human_creation.py:1-26: joints 0..41
        if body in (TOOL, TOOL2): return w.n_tool_links
        return 0
    p.getNumJoints = getNumJoints

    def getDynamicsInfo(body, link, physicsClientId=0):
        # (mass, lateral friction, ...): only what the state capture reads -- the ground's friction drawn by build_assistive_env (env.py:120)
        assert body == PLANE and link == -1, 'getDynamicsInfo: only the plane is backed'
        return (0.0, W().plane_friction, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), 0.0, 0.0, 0.0, -1.0, -1.0, 2, 0.001)
    p.getDynamicsInfo = getDynamicsInfo

    def getJointInfo(body, j, physicsClientId=0):
        w = W()
        if w is None:       # recording mode: the limits as createMultiBody received them
            b = RECORD['bodies'][-1]
            return (j, b'', b['linkJointTypes'][j], -1, -1, 0, 0.0, 0.0, float(b['linkLowerLimits'][j]), float(b['linkUpperLimits'][j]), 0.0, 0.0, b'', (0, 0, 1), (0, 0, 0), (0, 0, 0, 1), -1)
        d = w.dof_of.get((body, j))
        if d is None:
            jt, lo, hi, mf = p.JOINT_FIXED, 0.0, -1.0, 0.0
        else:
            jt = p.JOINT_PRISMATIC if w.blob.robot_i(d, 'JTYPE', w.gender) == 1 else p.JOINT_REVOLUTE
            lo, hi = w.dof_limits(d); mf = w.blob.robot_f(d, 'MAXF', gender=w.gender)
        return (j, b'joint%d' % j, jt, -1, -1, 0, 0.0, 0.0, lo, hi, mf, 0.0, b'link%d' % j, (0, 0, 1), (0, 0, 0), (0, 0, 0, 1), -1)
    p.getJointInfo = getJointInfo

    def getJointStates(body, jointIndices, physicsClientId=0):
        w = W()
        if w is None: return tuple((0.0, 0.0, (0.0,) * 6, 0.0) for _ in jointIndices)
        q, qd, _ = w.joints()
        out = []
        for j in jointIndices:
            d = w.dof_of.get((body, j))
            out.append((0.0, 0.0, (0.0,) * 6, 0.0) if d is None else (float(q[d]), float(qd[d]), (0.0,) * 6, 0.0))
        return tuple(out)
    p.getJointStates = getJointStates
    p.getJointState = lambda body, j, physicsClientId=0: getJointStates(body, [j])[0]

    def resetJointState(body, jointIndex, targetValue, targetVelocity=0, physicsClientId=0):
        w = W()
        if w is None: return
        d = w.dof_of.get((body, jointIndex))
        if d is None:
            w.ignored.append(('resetJointState', body, jointIndex, targetValue)); return
        if w.adopting:       # Agent.init's enforce_joint_limits (agent.py:24) is reset-time work that happened before this state was recorded
            w.adopt_resets.append((body, jointIndex, targetValue)); return
        w.L.agxo_world_reset_joint(w.h, d, float(targetValue), float(targetVelocity))
    p.resetJointState = resetJointState

    def setJointMotorControlArray(body, jointIndices, controlMode, targetPositions=None, positionGains=None, forces=None, physicsClientId=0, **k):
        w = W(); b = w.blob
        assert controlMode == p.POSITION_CONTROL
        for i, j in enumerate(jointIndices):
            d = w.dof_of.get((body, j))
            if d is None:
                w.ignored.append(('setJointMotorControlArray', body, j)); continue
            kp, mf = b.robot_f(d, 'KP', gender=w.gender), b.robot_f(d, 'MAXF', gender=w.gender)
            if d >= b.nrobot and w.human_kp > 0: kp, mf = w.human_kp, w.human_maxf
            if positionGains is not None and abs(float(positionGains[i]) - kp) > 1e-6 * max(1.0, abs(kp)): w.gain_mismatches.append((body, j, 'gain', float(positionGains[i]), kp))
            if forces is not None and abs(float(forces[i]) - mf) > 1e-6 * max(1.0, abs(mf)): w.gain_mismatches.append((body, j, 'force', float(forces[i]), mf))
            w.L.agxo_world_set_target(w.h, d, float(targetPositions[i]))
    p.setJointMotorControlArray = setJointMotorControlArray

    def stepSimulation(physicsClientId=0):
        w = W(); w.L.agxo_world_step(w.h); w.steps += 1
    p.stepSimulation = stepSimulation

    def getBasePositionAndOrientation(body, physicsClientId=0):
        pos, quat = W().link_frame(body, -1)
        return tuple(pos), tuple(quat)
    p.getBasePositionAndOrientation = getBasePositionAndOrientation

    def getBaseVelocity(body, physicsClientId=0):
        _, _, lin, ang = W().link_frame(body, -1, vel=True)
        return tuple(lin), tuple(ang)
    p.getBaseVelocity = getBaseVelocity

    def getLinkState(body, link, computeLinkVelocity=0, computeForwardKinematics=0, physicsClientId=0):
        pos, quat, lin, ang = W().link_frame(body, link, vel=True)
        # (comPos, comOrn, localInertialPos, localInertialOrn, linkFramePos, linkFrameOrn, linVel, angVel): the model keeps link frames only
        return (tuple(pos), tuple(quat), (0, 0, 0), (0, 0, 0, 1), tuple(pos), tuple(quat), tuple(lin), tuple(ang))
    p.getLinkState = getLinkState

    def resetBasePositionAndOrientation(body, pos, orn, physicsClientId=0):
        w = W()
        if body in w.markers:
            w.markers[body] = np.asarray(pos, dtype=np.float64).copy()
            if body == getattr(w, 'attach_marker', None):
                w.L.agxo_world_set_anchor(w.h, _p(np.ascontiguousarray(pos, dtype=np.float64)))
        elif WATER0 <= body < WATER0 + w.nwater:
            w.L.agxo_world_set_particle(w.h, body - WATER0, _p(np.ascontiguousarray(pos, dtype=np.float64)))
        elif FOOD0 <= body < FOOD0 + w.nfood:
            w.L.agxo_world_set_free_base(w.h, C.c_int(w.food0 + body - FOOD0), _p(np.ascontiguousarray(pos, dtype=np.float64)), _p(np.ascontiguousarray(orn, dtype=np.float64)))
        else:
            w.ignored.append(('resetBasePositionAndOrientation', body))
    p.resetBasePositionAndOrientation = resetBasePositionAndOrientation

    def _tuple(w, row, swap):
        ca, cb = int(row[0]), int(row[1])
        (ba, la), (bb, lb) = w.coll[ca], w.coll[cb]
        pa, pb, n = row[2:5], row[5:8], row[8:11]
        if swap: ba, la, bb, lb, pa, pb, n = bb, lb, ba, la, pb, pa, -n
        return (0, ba, bb, la, lb, tuple(pa), tuple(pb), tuple(n), float(row[11]), float(row[12]), 0.0, (0, 0, 0), 0.0, (0, 0, 0))

    def _water_tuple(w, body, other):
        pos = tuple(w.link_frame(body, -1)[0])
        return (0, body, other, -1, -1, pos, pos, (0.0, 0.0, 1.0), 0.0, 0.0, 0.0, (0, 0, 0), 0.0, (0, 0, 0))

    def getContactPoints(bodyA=None, bodyB=None, linkIndexA=None, linkIndexB=None, physicsClientId=0):
        w = W(); out = []
        if bodyA is not None and WATER0 <= bodyA < WATER0 + w.nwater:
            # a water particle: only WHETHER it touches the person is backed (drinking.py:85 reads the length of the list)
            assert bodyB == HUMAN, 'contacts of a water particle: only against the person'
            return (_water_tuple(w, bodyA, HUMAN),) if w.L.agxo_world_particle_query(w.h, bodyA - WATER0, 0.0) & 2 else ()
        for row in w.contacts():
            (ba, la), (bb, lb) = w.coll[int(row[0])], w.coll[int(row[1])]
            for swap in (False, True):
                if swap: ba, la, bb, lb = bb, lb, ba, la
                if bodyA is not None and ba != bodyA: continue
                if bodyB is not None and bb != bodyB: continue
                if linkIndexA is not None and la != linkIndexA: continue
                if linkIndexB is not None and lb != linkIndexB: continue
                out.append(_tuple(w, row, swap)); break
        return tuple(out)
    p.getContactPoints = getContactPoints

    def getClosestPoints(bodyA, bodyB, distance, linkIndexA=None, linkIndexB=None, physicsClientId=0):
        w = W()
        if WATER0 <= bodyA < WATER0 + w.nwater:
            assert bodyB == TOOL, 'closest points of a water particle: only against the cup (drinking.py:77)'
            return (_water_tuple(w, bodyA, TOOL),) if w.L.agxo_world_particle_query(w.h, bodyA - WATER0, float(distance)) & 1 else ()
        rows = w.closest(w.colliders_of(bodyA, linkIndexA), w.colliders_of(bodyB, linkIndexB), float(distance))
        out = []
        for r in rows:
            row = np.zeros(16); row[:8] = r[:8]; row[11] = r[8]
            out.append(_tuple(w, row, False))
        return tuple(out)
    p.getClosestPoints = getClosestPoints

    def getSoftBodyData(cloth, physicsClientId=0):
        w = W(); nn = w.oracle.L.agxo_cloth_nodes(C.c_void_p(w.oracle.h))
        x = np.zeros((nn, 3)); con = np.zeros((4 * nn, 6))
        n = w.L.agxo_world_cloth(w.h, _p(x), _p(con), C.c_int(len(con)))
        con = con[:max(n, 0)]
        return (x[:, 0], x[:, 1], x[:, 2], con[:, 0], con[:, 1], con[:, 2], con[:, 3], con[:, 4], con[:, 5])
    p.getSoftBodyData = getSoftBodyData
    return p


# ------------------------------------------------------------------------------------------------ inert stubs for the non-physics imports
class _Box:
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype), dtype
        self.shape = self.low.shape


class KerasLimitsModel:
    """what keras.models.load_model returns for assets/realistic_arm_limits_model.h5, as far as human.py:146 uses it:
    predict_classes of Sequential[Dense(64, tanh) x 3, Dense(1, sigmoid)].  Weights read from the reference's own file."""

    def __init__(self, path):
        sys.path.insert(0, ROOT)
        from assistive_gym_amd.model.h5lite import load_keras_dense_stack
        self.layers = [(np.asarray(k, dtype=np.float64), np.asarray(b, dtype=np.float64)) for k, b in load_keras_dense_stack(path)]

    def predict(self, x):
        h = np.asarray(x, dtype=np.float64)
        for i, (k, b) in enumerate(self.layers):
            h = h @ k + b.ravel()
            h = np.tanh(h) if i < len(self.layers) - 1 else 1.0 / (1.0 + np.exp(-h))
        return h

    def predict_classes(self, x):
        return (self.predict(x) > 0.5).astype(np.int32)


_CURRENT = [None]
RECORD = dict(shapes=[], bodies=[], filter=[])


def install():
    """Puts the facade and the inert stubs into sys.modules (idempotent) and returns the pybullet module object."""
    if 'pybullet' in sys.modules and getattr(sys.modules['pybullet'], '_agx_bridge', False):
        return sys.modules['pybullet']
    pb = make_pybullet(lambda: _CURRENT[0])
    pb._agx_bridge = True
    gym = types.ModuleType('gym'); spaces = types.ModuleType('gym.spaces'); utils = types.ModuleType('gym.utils'); seeding = types.ModuleType('gym.utils.seeding')
    gym.Env = type('Env', (), {})
    spaces.Box = _Box
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)       # gym 0.x: a RandomState (env.py:81)
    gym.spaces, gym.utils, utils.seeding = spaces, utils, seeding
    envs = types.ModuleType('gym.envs'); registration = types.ModuleType('gym.envs.registration'); registration.register = lambda **k: None
    gym.envs, envs.registration = envs, registration
    screeninfo = types.ModuleType('screeninfo'); screeninfo.get_monitors = lambda: []
    keras = types.ModuleType('keras'); kmodels = types.ModuleType('keras.models'); kmodels.load_model = lambda path: KerasLimitsModel(path); keras.models = kmodels
    ray = types.ModuleType('ray'); rllib = types.ModuleType('ray.rllib'); renv = types.ModuleType('ray.rllib.env'); mae = types.ModuleType('ray.rllib.env.multi_agent_env')
    mae.MultiAgentEnv = type('MultiAgentEnv', (), {})
    tune = types.ModuleType('ray.tune'); reg = types.ModuleType('ray.tune.registry'); reg.register_env = lambda name, creator: None
    ray.rllib, rllib.env, renv.multi_agent_env, ray.tune, tune.registry = rllib, renv, mae, tune, reg
    # <task>_envs.py imports the *Mesh env and agents.human_mesh at module level (smplx, trimesh: not installed, off the hot path)
    extra = {k: types.ModuleType(k) for k in ('smplx', 'trimesh') if k not in sys.modules}
    mods = {**extra, 'pybullet': pb, 'gym': gym, 'gym.spaces': spaces, 'gym.utils': utils, 'gym.utils.seeding': seeding, 'gym.envs': envs, 'gym.envs.registration': registration,
            'screeninfo': screeninfo, 'keras': keras, 'keras.models': kmodels, 'ray': ray, 'ray.rllib': rllib, 'ray.rllib.env': renv,
            'ray.rllib.env.multi_agent_env': mae, 'ray.tune': tune, 'ray.tune.registry': reg}
    for k, v in mods.items():
        sys.modules[k] = v
    return pb


def reference_envs():
    """imports /root/reference/assistive_gym/envs/*.py (unmodified) under the stubs; returns the `assistive_gym.envs` package.
    The package's __init__ imports gym registration and every task incl. the mesh / smplx ones; the task modules are imported one
    by one instead, under a private package name so that the drop-in `assistive_gym` of this repo is not shadowed."""
    install()
    name = '_agx_reference'
    if name in sys.modules:
        return sys.modules[name + '.envs']
    pkg = types.ModuleType(name); pkg.__path__ = [os.path.join(REF_ROOT, 'assistive_gym')]
    sub = types.ModuleType(name + '.envs'); sub.__path__ = [os.path.join(REF_ROOT, 'assistive_gym', 'envs')]
    sys.modules[name], sys.modules[name + '.envs'] = pkg, sub
    pkg.envs = sub
    for m in ('feeding_envs', 'bed_bathing_envs', 'scratch_itch_envs', 'dressing_envs', 'arm_manipulation_envs', 'drinking_envs'):
        setattr(sub, m, importlib.import_module('%s.envs.%s' % (name, m)))
    return sub


# ------------------------------------------------------------------------------------------------ adopting a state record
TASK_OF_KIND = {0: 'feeding', 1: 'bed_bathing', 2: 'scratch_itch', 3: 'dressing', 4: 'arm_manipulation', 5: 'drinking'}
ROBOT_CLASS = {'jaco': 'Jaco', 'sawyer': 'Sawyer', 'pr2': 'PR2', 'baxter': 'Baxter', 'panda': 'Panda', 'stretch': 'Stretch'}
TASK_CLASS = {'feeding': 'Feeding', 'bed_bathing': 'BedBathing', 'scratch_itch': 'ScratchItch', 'dressing': 'Dressing', 'arm_manipulation': 'ArmManipulation', 'drinking': 'Drinking'}


def adopt(blob, state, cloth=None):
    """Builds the reference's env object for `blob` (its own class, its own constructor) and wires it to a World holding `state`,
    the way the task's reset() leaves the object (citations inline).  Returns (env, world)."""
    envs = reference_envs()
    task = TASK_OF_KIND[blob.task_kind]
    robot = blob.meta.get('robot', 'jaco')
    cls = getattr(getattr(envs, task + '_envs'), '%s%s%sEnv' % (TASK_CLASS[task], ROBOT_CLASS[robot], 'Human' if blob.is_coop else ''))
    w = World(blob, state, cloth)
    _CURRENT[0] = w
    w.adopting = True
    env = cls()                                               # AssistiveEnv.__init__: config.ini, the arm-limit model, action / observation spaces
    from _agx_reference.envs.agents.agent import Agent
    from _agx_reference.envs.agents.robot import Robot
    from _agx_reference.envs.agents.tool import Tool
    v = blob.view(w.state0.reshape(1, -1))
    R, H = env.robot, env.human
    # AssistiveEnv.reset (env.py:93-112)
    env.agents, env.last_sim_time, env.iteration, env.forces, env.task_success = [], None, int(v['iteration'][0]), [], 0
    # build_assistive_env (env.py:114-134): robot.init -> Agent.init reads the joint limits from the engine
    links = [j for (b, j) in w.dof_of if b == ROBOT]
    ee = [R.right_end_effector, R.left_end_effector]
    w.n_robot_joints = max(links + ee + R.controllable_joint_indices) + 1
    w.n_tool_links = 0 if task in ('feeding', 'arm_manipulation', 'drinking') else 2      # spoon / scooper / cup: one body; wiper.urdf, tool_scratch.urdf: links 0, 1
    arm_right = task in ('feeding', 'arm_manipulation', 'drinking')            # feeding.py:142, drinking.py:150 arm='right'; bed_bathing.py:147, scratch_itch.py:116, dressing.py:134 arm='left'
    w.ee_links = {(R.right_end_effector if arm_right else R.left_end_effector): 0}
    if task == 'arm_manipulation' and not R.has_single_arm:
        w.ee_links[R.left_end_effector] = 1
    env.plane.body, env.plane.id = PLANE, env.id                                  # env.py:118
    R.body = ROBOT
    Robot.init(R, ROBOT, env.id, env.np_random)
    env.agents.append(R)
    # Human.init (human.py:72-102) minus create_human: gender, impairment and its draws come from the state record
    H.limits_model = env.human_limits_model
    H.arm_previous_valid_pose = {True: None, False: None}
    H.gender = 'male' if w.gender == 0 else 'female'
    nh = blob.nhdof
    trem = v['tremor'][0].astype(np.float64)
    H.impairment = 'tremor' if np.any(trem != 0) else ('limits' if w.limit_scale != 1.0 else 'none')
    H.limit_scale, H.strength = w.limit_scale, 1.0
    if float(v['human_kp'][0]) > 0:                                           # impairment 'weakness': the reactive hold's force is reactive_force x strength (human.py:126)
        H.strength = float(v['human_maxf'][0]) / {'scratch_itch': 1.0, 'dressing': 1.0, 'arm_manipulation': 0.01}.get(task, 1.0)
    hdofs = [d for d in range(blob.nrobot, blob.ndof)]
    pb_of = {blob.robot_i(d, 'PB_INDEX', w.gender): d for d in hdofs}
    assert all(j in pb_of for j in H.controllable_joint_indices), 'the model does not simulate every controllable joint of the human'
    H.tremors = np.array([trem[pb_of[j] - blob.nrobot] for j in H.controllable_joint_indices])
    rad = {0: 0.043, 1: 0.0355}[w.gender]                                      # human_creation.py:89,140
    H.hand_radius = H.elbow_radius = H.shoulder_radius = rad
    H.body = HUMAN
    Agent.init(H, HUMAN, env.id, env.np_random, H.controllable_joint_indices)
    if H.controllable or H.impairment == 'tremor':
        env.agents.append(H)                                                   # env.py:130-131
    # setup_joints (human.py:104-127): target_joint_angles = the tremor-free targets
    H.target_joint_angles = np.array([v['tremor_target'][0][pb_of[j] - blob.nrobot] for j in H.controllable_joint_indices], dtype=np.float64)
    if blob.task_i('ARM_LIMIT_ON') and int(v['task'][0][10]):                  # arm_previous_valid_pose (human.py:147-149)
        H.arm_previous_valid_pose[blob.task_f('ARM_LIMIT_SIGN') < 0] = [float(x) for x in w.state0[blob.h['S_TASK'] + 6:blob.h['S_TASK'] + 10]]
    # the tool(s)
    T = env.tool
    T.body, T.id, T.np_random = TOOL, env.id, env.np_random
    Agent.init(T, TOOL, env.id, env.np_random, indices=-1)
    if task == 'feeding':
        R.motor_gains = H.motor_gains = 0.025                                  # feeding.py:121
        env.generate_target()                                                  # feeding.py:184-196 (mouth offset by gender, update_targets)
        env.foods, env.foods_active = [], []
        alive, active = int(v['food_alive'][0]), int(v['food_active'][0])
        for k in range(blob.nfood):
            f = Agent(); f.init(FOOD0 + k, env.id, env.np_random, indices=-1)
            if alive >> k & 1: env.foods.append(f)
            if active >> k & 1: env.foods_active.append(f)
        env.total_food_count = int(v['total_food'][0])                         # feeding.py:169
        env.task_success = int(v['task_success'][0])
    elif task == 'drinking':
        R.motor_gains = H.motor_gains = 0.005                                  # drinking.py:130
        env.generate_target()                                                  # drinking.py:184-196
        env.cup_top_center_offset, env.cup_bottom_center_offset = np.array([0, 0, -0.055]), np.array([0, 0, 0.07])      # drinking.py:142-143
        tw = [int(x) & 0xffffffff for x in v['task'][0][:4]]                  # AGX_DK_ALIVE / AGX_DK_ACTIVE: 64-bit masks in two words each
        alive, active = tw[0] | tw[1] << 32, tw[2] | tw[3] << 32
        env.waters, env.waters_active = [], []
        for k in range(w.nwater):
            f = Agent(); f.init(WATER0 + k, env.id, env.np_random, indices=-1)
            if alive >> k & 1: env.waters.append(f)
            if active >> k & 1: env.waters_active.append(f)
        env.total_water_count = int(v['total_food'][0])                        # drinking.py:171
        env.task_success = int(v['task_success'][0])
    elif task == 'bed_bathing':
        w.first_target_marker = w.next_marker + 1
        env.generate_targets()                                                 # bed_bathing.py:173-203, the reference's own capsule_points
        alive = [int(x) & 0xffffffff for x in v['task'][0][:6]]
        nu = len(env.targets_pos_on_upperarm)
        keep_u = [i for i in range(nu) if alive[i >> 5] >> (i & 31) & 1]
        keep_f = [i for i in range(len(env.targets_pos_on_forearm)) if alive[(nu + i) >> 5] >> ((nu + i) & 31) & 1]
        for nm, keep in (('upperarm', keep_u), ('forearm', keep_f)):                # the lists get_total_force prunes (bed_bathing.py:62-74)
            for attr in ('targets_pos_on_' + nm, 'targets_' + nm, 'targets_pos_' + nm + '_world'):
                setattr(env, attr, [t for i, t in enumerate(getattr(env, attr)) if i in keep])
        env.task_success = int(v['task_success'][0])
    elif task == 'scratch_itch':
        env.prev_target_contact_pos = w.state0[blob.h['S_TASK'] + 12:blob.h['S_TASK'] + 15].astype(np.float64)     # scratch_itch.py:96
        env.limb = [H.right_shoulder, H.right_elbow][int(v['task'][0][3])]                                             # scratch_itch.py:137
        env.target_on_arm = w.state0[blob.h['S_TASK']:blob.h['S_TASK'] + 3].astype(np.float64)
        env.target = env.create_sphere(radius=0.01, mass=0.0, pos=[0, 0, 0], visual=True, collision=False)
        env.update_targets()
        env.task_success = int(v['task_success'][0])
    elif task == 'arm_manipulation':
        R.motor_forces, H.motor_forces = 20.0, 2.0                             # arm_manipulation.py:114-115
        if not R.has_single_arm:
            T2 = env.tool_left
            T2.body, T2.id, T2.np_random = TOOL2, env.id, env.np_random
            Agent.init(T2, TOOL2, env.id, env.np_random, indices=-1)
        env.task_success = float(w.state0[blob.h['S_TASK']])                   # AGX_AM_BEST
    elif task == 'dressing':
        R.motor_gains = H.motor_gains = 0.01                                   # dressing.py:118
        if R.mobile:
            R.gains = list(np.array(R.gains) / 8.0)                            # dressing.py:135-137: "Change robot gains since we use numSubSteps=8"
        env.cloth_forces = np.zeros((1, 1))                                    # dressing.py:113
        env.cloth = CLOTH
        env.triangle1_point_indices = [1180, 2819, 30]; env.triangle2_point_indices = [1322, 13, 696]   # dressing.py:156-157 (node numbering of the loaded mesh)
        tri = [int(x) for x in blob.i[blob.h['OFF_CLOTH'] + 13:blob.h['OFF_CLOTH'] + 19]]
        env.triangle1_point_indices, env.triangle2_point_indices = tri[:3], tri[3:]                   # the same six vertices in the blob's node order (AGX_CL_TRI)
        env.cloth_attachment = env.create_sphere(radius=0.0001, mass=0, pos=[0, 0, 0], visual=True, collision=False)
        w.attach_marker = env.cloth_attachment.body
        env.task_success = float(w.state0[blob.h['S_TASK'] + 2])               # AGX_DR_BEST
        env.update_targets()                                                   # dressing.py:200-210: attachment at the end effector
    env.init_env_variables = lambda reset=False: None
    w.adopting = False
    return env, w


def split_action(env, action):
    """the action as the env's step() takes it: a dict in co-op (feeding.py:13-14)"""
    a = np.asarray(action, dtype=np.float32)
    if env.human.controllable:
        n = len(env.robot.controllable_joint_indices)
        return {'robot': a[:n].copy(), 'human': a[n:].copy()}
    return a.copy()


def flat_obs(obs):
    return np.concatenate([obs['robot'], obs['human']]) if isinstance(obs, dict) else np.asarray(obs)


def ref_step(blob, state, action, cloth=None):
    """one env.step() of the reference's class on the oracle's physics.  Returns dict(obs, reward, done, info, state, cloth, world)."""
    env, w = adopt(blob, state, cloth)
    obs, rew, done, info = env.step(split_action(env, action))
    if isinstance(rew, dict):
        rew, done, info = rew['robot'], done['__all__'], info['robot']
    st, cl = w.store()
    st = writeback(env, w, st)
    out = dict(obs=flat_obs(obs).astype(np.float64), reward=float(rew), done=bool(done), info=info, state=st, cloth=cl, env=env, world=w)
    return out


def writeback(env, w, state):
    """The bookkeeping the reference keeps in Python attributes, written into the state record the way the stepper stores it, so
    that the record after a reference step can be compared word by word with the oracle's / the device's."""
    blob = w.blob
    v = blob.view(state.reshape(1, -1))
    task = TASK_OF_KIND[blob.task_kind]
    H = env.human
    v['iteration'][0] = env.iteration
    pb_of = {blob.robot_i(d, 'PB_INDEX', w.gender): d for d in range(blob.nrobot, blob.ndof)}
    if H.target_joint_angles is not None and H in env.agents and H.impairment == 'tremor':
        for j, t in zip(H.controllable_joint_indices, H.target_joint_angles):
            v['tremor_target'][0][pb_of[j] - blob.nrobot] = t
    st = blob.h['S_TASK']
    right = blob.task_f('ARM_LIMIT_SIGN') < 0                                   # human.py:138-145: -1 right arm, +1 left arm
    if blob.task_i('ARM_LIMIT_ON') and H.arm_previous_valid_pose[right] is not None:
        state[st + 6:st + 10] = np.asarray(H.arm_previous_valid_pose[right], dtype=np.float32)
        state.view(np.int32)[st + 10] = 1
    if task == 'feeding':
        v['target'][0] = env.target_pos
        alive = sum(1 << (f.body - FOOD0) for f in env.foods); active = sum(1 << (f.body - FOOD0) for f in env.foods_active)
        v['food_alive'][0], v['food_active'][0], v['task_success'][0] = alive, active, env.task_success
    elif task == 'drinking':
        v['target'][0] = env.target_pos
        alive = sum(1 << (f.body - WATER0) for f in env.waters); active = sum(1 << (f.body - WATER0) for f in env.waters_active)
        state.view(np.uint32)[st:st + 4] = np.array([alive & 0xffffffff, alive >> 32, active & 0xffffffff, active >> 32], dtype=np.uint32)
        v['task_success'][0] = env.task_success
    elif task == 'bed_bathing':
        v['task_success'][0] = env.task_success
        # the surviving targets by identity: the marker bodies were created upper arm first, in order (adopt -> generate_targets)
        alive = [0] * 6
        ids = sorted(m for m in w.markers)
        first = {m: i for i, m in enumerate(ids)}
        for t in list(env.targets_upperarm) + list(env.targets_forearm):
            i = first[t.body] - first[w.first_target_marker]
            alive[i >> 5] |= 1 << (i & 31)
        state.view(np.uint32)[st:st + 6] = np.array(alive, dtype=np.uint32)
    elif task == 'scratch_itch':
        v['task_success'][0] = env.task_success
        state[st + 12:st + 15] = np.asarray(env.prev_target_contact_pos, dtype=np.float32)
    elif task == 'arm_manipulation':
        state[st] = np.float32(env.task_success)
    elif task == 'dressing':
        state[st + 1] = np.float32(env.cloth_force_sum)
        state[st + 2] = np.float32(env.task_success)
    return state


def compare_states(blob, a, b):
    """max deviations between two state records, by section: dict(q, qd, qt, free_pos, free_vel, ints_equal, task)"""
    va, vb = blob.view(a.reshape(1, -1).copy()), blob.view(b.reshape(1, -1).copy())
    out = {k: float(np.abs(va[k].astype(np.float64) - vb[k].astype(np.float64)).max()) if va[k].size else 0.0 for k in ('q', 'qd', 'qt', 'tremor_target', 'target')}
    fa, fb = va['free'][0].astype(np.float64), vb['free'][0].astype(np.float64)
    near = np.abs(fa[:, :3]).max(axis=1) < 500 if len(fa) else np.zeros(0, bool)      # eaten particles are teleported to random far-away places
    out['free_pos'] = float(np.abs(fa[near, :7] - fb[near, :7]).max()) if near.any() else 0.0
    out['free_vel'] = float(np.abs(fa[near, 7:] - fb[near, 7:]).max()) if near.any() else 0.0
    out['ints_equal'] = all(int(va[k][0]) == int(vb[k][0]) for k in ('food_alive', 'food_active', 'iteration', 'task_success', 'gender', 'frozen'))
    out['task_words'] = (va['task'][0], vb['task'][0])
    return out
