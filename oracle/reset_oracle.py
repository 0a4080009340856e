"""TEST INFRASTRUCTURE ONLY -- numpy float64 restatement of the device-side reset generator
(assistive_gym_amd/csrc/agx_reset.h).  Only tests/ may import this file; the product never does.

What it restates, in the order of FeedingEnv.reset (assistive_gym/envs/feeding.py:114-182):
  plane friction U(0.025, 0.5)                         envs/env.py:120
  gender, impairment, limit scale, strength, tremors   envs/agents/human.py:72-92
  human pose: presets + head angles U(-30, 30) deg     feeding.py:124-125, human.py:104-127 (clamped to the limits,
                                                       agents/agent.py:240-250), tree of human_creation.py:188-278
  mouth target                                         feeding.py:184-196
  end-effector start position + IK with random restarts  feeding.py:139, env.py:276-310, robot.py:84-121
  gripper, tool in the hand, bowl offset, food grid    feeding.py:143-166, tool.py:49-62, furniture.py:32-34

PARITY UNPINNED with respect to PyBullet: the IK is this repository's damped least squares (Bullet's
calculateInverseKinematics is not reproducible) and the random stream is a counter-based
Philox4x32-10 instead of numpy's MT19937 (the reference's draw COUNT depends on Bullet's IK results), so
this file pins the DEVICE KERNEL to an independent restatement, not to the reference's seeds.

Random numbers: u(stream, idx) = Philox4x32-10(counter = (idx, stream, 0, 0), key = 64-bit env seed), first two
output words -> 53-bit double in [0, 1).  Stream 0 holds the scalar draws of one reset (slots below), stream
1 + r the draws of IK restart r.  Every draw has a fixed slot, so restarts can be evaluated in any order
(the device evaluates 64 at a time, one per lane) and "the first successful restart" stays well defined.
"""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF
# stream 0 slots
S_FRICTION, S_GENDER, S_IMPAIRMENT, S_LIMIT, S_STRENGTH, S_HEAD, S_EE, S_BOWL, S_TREMOR = 0, 1, 2, 3, 4, 8, 12, 16, 32
S_LIMB, S_TARGET_LEN, S_TARGET_TH = 48, 49, 50          # scratch itch: generate_target (scratch_itch.py:134-146)
S_RAGDOLL = 64                                           # + k: the jitter of the rag doll's k-th joint (bed_bathing.py:126)
# restart stream slots (+ DoF index)
R_REST, R_LO, R_HI = 0, 16, 32
# base pose search (free-standing robots): stream T_STREAM0 + 64 (placement x rounds + round) + candidate
T_STREAM0, T_X, T_Y, T_YAW, T_REST = 2000, 0, 1, 2, 16
IMPAIRMENTS = ('none', 'limits', 'weakness', 'tremor')       # human.py:80
MODE_RANDOM, MODE_NO_TREMOR = -1, -2

# layout tables of the reset section (include/agx_blob.h AGX_X_*, AGX_XJ_*); restated here so that the
# oracle does not depend on the product's Python package
X_ = dict(NJOINT=0, NARM=1, BASE_POS=2, BASE_QUAT=5, EE_QUAT=9, EE_TARGET=13, EE_RANGE=16, BOWL_POS=17, BOWL_RANGE=20,
          HBASE_M=21, HBASE_F=24, FOOD_R=27, HEAD_RANGE=28, IK_ITERS=29, IK_DAMP=30, IK_MAXSTEP=31, IK_THRESH=32,
          IK_RESTARTS=33, IK_TOL=34, IK_RANDLIM_FROM=35, FRIC_LO=36, FRIC_HI=37, LIMIT_LO=38, TREMOR_RANGE=39,
          BOWL_BODY=40, OFF_JOINTS=41, OFF_BODIES=42, OFF_DYN=43, STRENGTH_LO=44, FOOD_OFF=45, COLLISION_TRIES=48,
          REACTIVE_KP=49, REACTIVE_MAXF=50, FLAGS=51, TOC_ATTEMPTS=52, TOC_ROUNDS=53, TOC_POS_RANGE=54, TOC_YAW_RANGE=55, TOC_YAW0=56, TOC_X_SIGN=57,
          TOC_IK_ITERS=58, TOC_THRESH=59, TOC_GOAL_LINKS=60, TOC_GOAL_ORIENT=63, TOC_GOAL_OFF=64, CLOTH_GRAVITY_SETTLE=67, CLOTH_GRAVITY=68,
          CLOTH_ORIG_POS=69, TOC_GOAL_QUAT=72, CHAIN=84, TOC_NGOALS=91, TOC_GOAL_KIND=92, PED_N=93, MOBILE_LIFT=94, MOBILE_LIFT_DOF=95, PED_BOX=96, TOC_GOAL_LINK3=108, FALL_PARK=109, CHAIN2=112, EE_TARGET2=119, COUNT=124)
XJ = dict(PARENT=0, OFF=1, AXIS=4, LOWER=7, UPPER=8, FLAGS=9, PRESET=10, DRAW=11, STRIDE=12)
H_OFF_RESET, H_OFF_ROBOT, H_OFF_FREE, H_OFF_TASK, H_S_TASK = 31, 13, 14, 18, 36
H_TASK_KIND, H_OFF_CLOTH = 35, 40
R = dict(PARENT=0, TPOS=1, TQUAT=4, AXIS=8, LOWER=21, UPPER=22, ACT=27, QT0=28, JTYPE=32, STRIDE=36)
F = dict(REFPOS=5, REFQUAT=8, STRIDE=16)
T = dict(SI_LIMB_DIMS=9, MOUTH_M=11, MOUTH_F=14, HEAD_LINK=17, EE_LINK=18, EE_POS=19, EE_QUAT=22, TOOL_POS=26, TOOL_QUAT=29, COOP=35,
         TOOL2_BODY=72, EE2_LINK=73, EE2_POS=74, EE2_QUAT=77, TOOL2_POS=81, TOOL2_QUAT=84)
S_EE2 = 20                                               # stream 0: the second arm's start position (arm_manipulation.py:159)


def contacts_collide(blob_words, contacts):
    """The verdict of reset_collides (csrc/agx_reset.h) on the C oracle's contact rows [colliderA, colliderB, pA, pB, n, distance, ...]:
    a robot link or the tool touching (distance <= 0) the human, the table or the wheelchair."""
    i = np.ascontiguousarray(blob_words, dtype=np.uint32).view(np.int32)
    off, stride, tag = int(i[15]), 16, 5                                   # AGX_H_OFF_COLL, AGX_C_STRIDE, AGX_C_TAG
    for c in contacts:
        ta, tb = int(i[off + int(c[0]) * stride + tag]), int(i[off + int(c[1]) * stride + tag])
        ra, rb = ta in (1, 2), tb in (1, 2)                                # AGX_TAG_ROBOT, AGX_TAG_TOOL
        oa, ob = ta in (3, 6, 8), tb in (3, 6, 8)                          # AGX_TAG_HUMAN, AGX_TAG_TABLE, AGX_TAG_WHEELCHAIR
        if ((ra and ob) or (rb and oa)) and c[11] <= 0:
            return True
    return False


def with_collision_check(blob_words):
    """ResetOracle whose collision verdict comes from the contact list of the C oracle (oracle/agx_oracle.c) for the sampled state"""
    import ctypes as C
    import os
    w = np.ascontiguousarray(blob_words, dtype=np.uint32)
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libagx_oracle.so'))
    lib.agxo_load.restype = C.c_void_p
    lib.agxo_load.argtypes = [C.c_void_p, C.c_size_t]
    lib.agxo_substep_debug.restype = C.c_int
    h = lib.agxo_load(w.ctypes.data_as(C.c_void_p), C.c_size_t(len(w)))
    assert h, 'oracle rejected the model blob'

    def collides(st):
        out = np.zeros((96, 13))
        s = np.ascontiguousarray(st, dtype=np.float32).copy()
        n = lib.agxo_substep_debug(C.c_void_p(h), s.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(96))
        return contacts_collide(w, out[:n])
    o = ResetOracle(w, collides)
    o._keep = (lib, w)
    return o


def philox4x32(counter, key):
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def u01(seed, stream, idx):
    c = philox4x32((idx & MASK, stream & MASK, 0, 0), (seed & MASK, (seed >> 32) & MASK))
    return (((c[0] >> 5) << 26) | (c[1] >> 6)) / 9007199254740992.0


# ---- float64 rigid transforms, (x, y, z, w) quaternions ------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qrot(q, v):
    x, y, z, w = q / np.sqrt(np.dot(q, q))
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return Rm @ v


def q_axis_angle(axis, angle):
    n = np.sqrt(np.dot(axis, axis))
    if n < 1e-12:
        return np.array([0, 0, 0, 1.0])
    s = np.sin(0.5 * angle) / n
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(0.5 * angle)])


def compose(pa, qa, pb, qb):
    return pa + qrot(qa, pb), qmul(qa, qb)


class ResetOracle:
    def __init__(self, words, collides=None):
        """collides: optional callback state record -> bool, see contacts_collide() (the C oracle's contact list)"""
        self.collides = collides
        w = np.ascontiguousarray(words, dtype=np.uint32)
        self.f = w.view(np.float32).astype(np.float64)          # every constant is the blob's float32 value, widened
        self.i = w.view(np.int32)
        self.x0 = int(self.i[H_OFF_RESET])
        self.nrobot, self.nhdof, self.ndof = int(self.i[32]), int(self.i[33]), int(self.i[3])
        self.nfree, self.nhuman, self.nfood, self.food0, self.tool_body = int(self.i[4]), int(self.i[5]), int(self.i[9]), int(self.i[27]), int(self.i[28])
        self.state_words = int(self.i[19])
        self.S = {k: int(self.i[v]) for k, v in dict(Q=20, QD=21, QT=22, FREE=23, BASE=24, HUMAN=25, ENV=26, TREMOR=34).items()}

    # -- accessors ---------------------------------------------------------------------------------
    def xf(self, key, n=1):
        o = self.x0 + X_[key]
        return self.f[o:o + n].copy() if n > 1 else float(self.f[o])

    def xi(self, key):
        return int(self.i[self.x0 + X_[key]])

    def jf(self, g, j, key, n=1):
        o = self.x0 + self.xi('OFF_JOINTS') + (g * self.xi('NJOINT') + j) * XJ['STRIDE'] + XJ[key]
        return self.f[o:o + n].copy() if n > 1 else float(self.f[o])

    def ji(self, g, j, key):
        return int(self.i[self.x0 + self.xi('OFF_JOINTS') + (g * self.xi('NJOINT') + j) * XJ['STRIDE'] + XJ[key]])

    def rf(self, d, key, n=1):
        o = int(self.i[H_OFF_ROBOT]) + d * R['STRIDE'] + R[key]
        return self.f[o:o + n].copy() if n > 1 else float(self.f[o])

    def ri(self, d, key):
        return int(self.i[int(self.i[H_OFF_ROBOT]) + d * R['STRIDE'] + R[key]])

    def tf(self, key, n=1):
        o = int(self.i[H_OFF_TASK]) + T[key]
        return self.f[o:o + n].copy() if n > 1 else float(self.f[o])

    def ti(self, key):
        return int(self.i[int(self.i[H_OFF_TASK]) + T[key]])

    # -- human --------------------------------------------------------------------------------------
    def joint_angle(self, g, j, ls, head):
        """task preset (+ head draw), clamped to the joint limits (agent.py:240-250); fixed joints are 0"""
        flags = self.ji(g, j, 'FLAGS')
        if not flags & 1:
            return 0.0
        if getattr(self, 'fell', None) is not None:                        # a joint of the arm that fell: where it came to rest (arm_manipulation.py:145-151)
            dyn = [int(self.i[self.x0 + self.xi('OFF_DYN') + k]) for k in range(self.nhdof)]
            if j in dyn:
                return float(self.fell[self.S['Q'] + self.nrobot + dyn.index(j)])
        fall_stage = bool(self.xi('FLAGS') & 256)
        if getattr(self, 'settled', None) is not None and not (fall_stage and flags & 4):     # where the rag doll came to rest (bed_bathing.py:129-137)
            a = float(self.settled[6 + j - (1 if j > 24 else 0)])
            if fall_stage or getattr(self, 'fell', None) is not None:      # setup_joints -> enforce_joint_limits (arm_manipulation.py:139-140, human.py:121); the task stage reads the SAME clamped pose the fall model wrote its bodies from
                s = ls if flags & 2 else 1.0
                a = min(max(a, self.jf(g, j, 'LOWER') * s), self.jf(g, j, 'UPPER') * s)
            return a
        a = self.jf(g, j, 'PRESET')
        k = self.ji(g, j, 'DRAW')
        if k >= 0:
            a = a + head[k]
        s = ls if flags & 2 else 1.0
        return min(max(a, self.jf(g, j, 'LOWER') * s), self.jf(g, j, 'UPPER') * s)

    def link_pose(self, g, link, ls, head):
        """world pose of a human link frame: walk from the link to the base, then the base transform"""
        p, q = np.zeros(3), np.array([0, 0, 0, 1.0])
        j = link
        while j >= 0:
            jq = q_axis_angle(self.jf(g, j, 'AXIS', 3), self.joint_angle(g, j, ls, head))
            p, q = compose(self.jf(g, j, 'OFF', 3), jq, p, q)
            j = self.ji(g, j, 'PARENT')
        if getattr(self, 'settled', None) is not None:                     # the rag doll's base: position q[0..2], orientation Rz(q[3]) Ry(q[4]) Rx(q[5])
            t = self.settled
            bq = qmul(qmul(q_axis_angle(np.array([0, 0, 1.0]), t[3]), q_axis_angle(np.array([0, 1.0, 0]), t[4])), q_axis_angle(np.array([1.0, 0, 0]), t[5]))
            return compose(t[:3], bq, p, q)
        return compose(self.xf('HBASE_F' if g else 'HBASE_M', 3), np.array([0, 0, 0, 1.0]), p, q)

    # -- robot --------------------------------------------------------------------------------------
    def chain(self):
        """DoFs of the arm's joints in chain order (AGX_X_CHAIN; self.arm = 1: the second arm of a two-armed robot, AGX_X_CHAIN2)"""
        return [int(self.i[self.x0 + X_['CHAIN2' if getattr(self, 'arm', 0) else 'CHAIN'] + k]) for k in range(self.xi('NARM'))]

    def arm_limits(self):
        ch = self.chain()
        return (np.array([self.rf(d, 'LOWER') for d in ch], dtype=np.float64), np.array([self.rf(d, 'UPPER') for d in ch], dtype=np.float64))

    def mobile_fk(self, lift_dof, lift_q, base):
        """end-effector pose of a robot on wheels (FLAGS bit 3; csrc/agx_reset.h rs_mobile_fk): the chain from the link carrying the end
        effector up to the base, every joint at its clamped QT0 but the lift"""
        links, d = [], self.ti('EE_LINK')
        while d >= 0:
            links.append(d)
            d = self.ri(d, 'PARENT')
        pp, pq = base
        for d in links[::-1]:
            jp, jq = compose(pp, pq, self.rf(d, 'TPOS', 3), self.rf(d, 'TQUAT', 4))
            qd = lift_q if d == lift_dof else min(max(self.rf(d, 'QT0'), self.rf(d, 'LOWER')), self.rf(d, 'UPPER'))
            ax = self.rf(d, 'AXIS', 3).astype(np.float64)
            if self.ri(d, 'JTYPE') == 1:
                pp, pq = jp + qrot(jq, ax * qd), jq
            else:
                pp, pq = jp, qmul(jq, q_axis_angle(ax, qd))
        return compose(pp, pq, self.tf('EE_POS', 3), self.tf('EE_QUAT', 4))

    def arm_fk(self, q, base=None):
        narm = self.xi('NARM')
        pp, pq = base if base is not None else (self.xf('BASE_POS', 3), self.xf('BASE_QUAT', 4))
        pos, axw = [], []
        ch = self.chain()
        for k in range(narm):
            d = ch[k]
            dup = int(self.i[int(self.i[H_OFF_TASK]) + 70]) if int(self.i[H_TASK_KIND]) == 4 else 0     # AGX_T_DUP_ACT: robot_arm = 'both' lists a single arm twice, the second copy's actions drive it (robot.py:16)
            assert self.ri(d, 'PARENT') == (ch[k - 1] if k else -1) and self.ri(d, 'ACT') == k + dup + (narm if getattr(self, 'arm', 0) else 0)
            jp, jq = compose(pp, pq, self.rf(d, 'TPOS', 3), self.rf(d, 'TQUAT', 4))
            ax = self.rf(d, 'AXIS', 3)
            pq = qmul(jq, q_axis_angle(ax, q[k]))
            pp = jp
            pos.append(jp)
            axw.append(qrot(pq, ax))
        second = getattr(self, 'arm', 0)
        assert self.ti('EE2_LINK' if second else 'EE_LINK') == ch[narm - 1]
        pe, oe = compose(pp, pq, self.tf('EE2_POS' if second else 'EE_POS', 3), self.tf('EE2_QUAT' if second else 'EE_QUAT', 4))
        return pe, oe, pos, axw

    def ik(self, q0, lo, hi, target_pos, target_quat, iters=None, base=None):
        """damped least squares; target_quat None = position only"""
        narm = self.xi('NARM')
        q = np.array(q0, dtype=np.float64)
        lam2 = self.xf('IK_DAMP') ** 2
        tol, maxstep = self.xf('IK_TOL'), self.xf('IK_MAXSTEP')
        for _ in range(self.xi('IK_ITERS') if iters is None else iters):
            pe, oe, pos, axw = self.arm_fk(q, base)
            ep = target_pos - pe
            er = np.zeros(3)
            if target_quat is not None:
                qe = qmul(target_quat, np.array([-oe[0], -oe[1], -oe[2], oe[3]]))
                if qe[3] < 0:
                    qe = -qe
                er = 2.0 * qe[:3]
            if np.sqrt(ep @ ep) < tol and np.sqrt(er @ er) < tol:
                break
            J = np.zeros((6, narm))
            for d in range(narm):
                J[:3, d] = np.cross(axw[d], pe - pos[d])
                J[3:, d] = axw[d]
            if target_quat is None:
                J, e = J[:3], ep
            else:
                e = np.concatenate([ep, er])
            y = np.linalg.solve(J @ J.T + lam2 * np.eye(len(e)), e)
            dq = J.T @ y
            step = np.max(np.abs(dq))
            if step > maxstep:
                dq = dq * (maxstep / step)
            q = np.minimum(np.maximum(q + dq, lo), hi)
        return q

    def jlwki(self, q, base):
        """joint-limit-weighted kinematic isotropy of an arm pose (robot.py:186-191, 217-228)"""
        narm = self.xi('NARM')
        pe, oe, pos, axw = self.arm_fk(q, base)
        J = np.zeros((6, narm))
        for d in range(narm):
            J[:3, d] = np.cross(axw[d], pe - pos[d])
            J[3:, d] = axw[d]
        lower, upper = self.arm_limits()
        qr = 0.5 * (upper - lower)
        w = np.maximum(1.0 - np.power(0.5, (qr - np.abs(qr - q + lower)) / (0.05 * qr) + 1.0), 0.001)
        M = (J * w[None]) @ J.T
        det = max(np.linalg.det(M), 0.0)
        return det ** (1.0 / 6.0) / (np.trace(M) / 6.0)

    def toc(self, seed, placement, target_pos, target_quat, goals, goal_quats=None, must=1, target_pos2=None):
        """Robot.position_robot_toc (robot.py:123-215) as the device runs it: TOC_ATTEMPTS candidate base poses per round, each solving
        the start pose and the position goals from random rest poses; -> (ok, base (p, q), start solution, rounds used, goals reached).
        target_pos2 (a two-armed robot, FLAGS bit 9; arm_manipulation.py:165): the second arm's start position -- arm 0 takes target_pos and
        the first half of the goals, arm 1 (CHAIN2, EE2) target_pos2 and the second half; both start poses must be reached; the start
        solution is then the pair (q of arm 0, q of arm 1)"""
        narm, A, rounds = self.xi('NARM'), self.xi('TOC_ATTEMPTS'), self.xi('TOC_ROUNDS')
        thr, pr, yr = self.xf('TOC_THRESH'), self.xf('TOC_POS_RANGE'), self.xf('TOC_YAW_RANGE')
        base0 = self.xf('BASE_POS', 3)
        two = target_pos2 is not None
        narms, gpa = (2, len(goals) // 2) if two else (1, len(goals))
        if two:
            must = 9
        out = None
        for rnd in range(rounds):
            best = None
            for a in range(A):
                stream = T_STREAM0 + 64 * (placement * rounds + rnd) + a
                yaw = self.xf('TOC_YAW0') + (2 * u01(seed, stream, T_YAW) - 1) * yr
                base = (base0 + np.array([self.xf('TOC_X_SIGN') * pr * u01(seed, stream, T_X), (2 * u01(seed, stream, T_Y) - 1) * pr, 0.0]),
                        q_axis_angle(np.array([0, 0, 1.0]), yaw))
                reached, manip, qs = 0, 0.0, [None, None]
                for arm in range(narms):
                    self.arm = arm
                    lower, upper = self.arm_limits()
                    lo, hi = np.where(lower < -1e9, -2 * np.pi, lower), np.where(upper > 1e9, 2 * np.pi, upper)
                    for g in range(1 + gpa):
                        slot = (3 * arm if two else 0) + g
                        q0 = lo + (hi - lo) * np.array([u01(seed, stream, T_REST + 8 * slot + d) for d in range(narm)])
                        tp = (target_pos2 if arm else target_pos) if g == 0 else goals[arm * gpa + g - 1]
                        tq = target_quat if g == 0 else (goal_quats[g - 1] if goal_quats is not None else None)
                        q = self.ik(q0, lo, hi, tp, tq, iters=self.xi('TOC_IK_ITERS'), base=base)
                        pe, oe, orig, _ = self.arm_fk(q, base)
                        hit = np.sqrt((tp - pe) @ (tp - pe)) < thr
                        if g == 0 and hit and self.xi('PED_N') > 0:                # the pedestal guard (host/reset_bed.py::_arm_in_pedestal)
                            pts = orig[2:] + [0.5 * (orig[k] + orig[k + 1]) for k in range(2, narm - 1)] + [pe]
                            bi = np.array([-base[1][0], -base[1][1], -base[1][2], base[1][3]])
                            for b in range(self.xi('PED_N')):
                                bx = self.f[self.x0 + X_['PED_BOX'] + 6 * b:self.x0 + X_['PED_BOX'] + 6 * b + 6].astype(np.float64)
                                for pt in pts:
                                    l = qrot(bi, pt - base[0])
                                    if np.all(l >= bx[:3]) and np.all(l <= bx[3:]):
                                        hit = False
                        if tq is not None:
                            dm, dp = tq - oe, tq + oe
                            hit = hit and min(np.sqrt(dm @ dm), np.sqrt(dp @ dp)) < thr
                        if g == 0:
                            qs[arm] = q
                        if hit:
                            reached |= 1 << slot
                            manip += self.jlwki(q, base)
                self.arm = 0
                ngoal = bin(reached).count('1') if (reached & must) == must else -1      # the start goals must be reachable (robot.py:196-200)
                if best is None or ngoal > best[0] or (ngoal == best[0] and ngoal > 0 and manip > best[1]):
                    best = (ngoal, manip, base, (qs[0], qs[1]) if two else qs[0])
            out = (best[0] > 0, best[2], best[3], rnd + 1, best[0], best[1])
            if best[0] > 0:
                break
        return out

    def restart(self, seed, r, target_pos, target_quat):
        """IK restart r (robot.py:88-99): returns (q, position error, orientation error)"""
        narm = self.xi('NARM')
        lower, upper = self.arm_limits()
        ik_lo = np.where(lower < -1e9, -2 * np.pi, lower)                 # agent.py:223-231
        ik_hi = np.where(upper > 1e9, 2 * np.pi, upper)
        lo, hi = ik_lo, ik_hi
        if r >= self.xi('IK_RANDLIM_FROM'):                               # robot.py:91 randomize_limits
            lo = np.array([u01(seed, 1 + r, R_LO + d) for d in range(narm)]) * ik_lo
            hi = np.array([u01(seed, 1 + r, R_HI + d) for d in range(narm)]) * ik_hi
        rest = lo + (hi - lo) * np.array([u01(seed, 1 + r, R_REST + d) for d in range(narm)])     # agent.py:263
        q = self.ik(rest, np.minimum(lo, hi), np.maximum(lo, hi), target_pos, target_quat)
        q = np.minimum(np.maximum(q, lower), upper)                       # set_joint_angles(use_limits=True)
        pe, oe, _, _ = self.arm_fk(q)
        dpos = np.sqrt((target_pos - pe) @ (target_pos - pe))
        dm, dp = target_quat - oe, target_quat + oe
        return q, dpos, min(np.sqrt(dm @ dm), np.sqrt(dp @ dp))

    # -- one reset ----------------------------------------------------------------------------------
    def ragdoll_drop(self, seed, impairment_mode=MODE_RANDOM, gender_mode=-1):
        """the sampler of the rag-doll model (FLAGS bit 5; bed_bathing.py:119-127): its drop record -- base in the air, every joint
        U(-r, r) clamped to its limits, at rest; friction / gender / limit scale as the task blob's sampler draws them from the same seed"""
        assert self.xi('FLAGS') & 32
        u = lambda idx: u01(seed, 0, idx)
        friction = self.xf('FRIC_LO') + (self.xf('FRIC_HI') - self.xf('FRIC_LO')) * u(S_FRICTION)
        g = gender_mode if gender_mode >= 0 else (0 if u(S_GENDER) < 0.5 else 1)
        if impairment_mode >= 0:
            imp = impairment_mode
        else:
            nchoice = 4 if impairment_mode == MODE_RANDOM else 3
            imp = min(int(u(S_IMPAIRMENT) * nchoice), nchoice - 1)
        ls = 1.0 if imp != 1 else self.xf('LIMIT_LO') + (1.0 - self.xf('LIMIT_LO')) * u(S_LIMIT)
        st = np.zeros(self.state_words, dtype=np.float32)
        si = st.view(np.int32)
        S = self.S
        q = np.zeros(self.ndof)
        q[:3] = self.xf('HBASE_F' if g else 'HBASE_M', 3)
        q[3:6] = self.xf('EE_TARGET', 3)
        r = self.xf('EE_RANGE')
        for k in range(self.ndof - 6):
            j = k + (1 if k >= 24 else 0)
            b0 = self.x0 + self.xi('OFF_JOINTS') + (g * self.xi('NJOINT') + j) * XJ['STRIDE']
            sc = ls if int(self.i[b0 + XJ['FLAGS']]) & 2 else 1.0
            q[6 + k] = min(max((2 * u(S_RAGDOLL + k) - 1) * r, float(self.f[b0 + XJ['LOWER']]) * sc), float(self.f[b0 + XJ['UPPER']]) * sc)
        st[S['Q']:S['Q'] + self.ndof] = q
        st[S['QT']:S['QT'] + self.ndof] = q
        st[S['HUMAN'] + 6] = 1.0
        st[S['BASE'] + 6] = 1.0
        e = S['ENV']
        st[e + 0], si[e + 1], st[e + 13] = friction, g, ls
        return st, dict(gender=g, impairment=imp, limit_scale=ls)

    def sample(self, seed, impairment_mode=MODE_RANDOM, gender_mode=-1, max_restarts=None, settled=None, fell=None):
        """-> (state record float32[state_words], info dict).  With a `collides` callback (state record -> bool: does the arm /
        tool touch the human, the table or the wheelchair?) a successful IK restart that collides is rejected and the search
        goes on from the next restart (robot.py:105-112, env.py:299-308), at most COLLISION_TRIES times."""
        first, rejected = 0, []
        tries = self.xi('COLLISION_TRIES') if self.collides is not None else 0
        for t in range(tries + 1):
            st, info = self._sample_from(seed, impairment_mode, gender_mode, max_restarts, first, settled, fell)
            if t == tries or not info['ik_ok'] or not self.collides(st):
                break
            placed = self.xi('TOC_ATTEMPTS') > 0 or bool(self.xi('FLAGS') & 8)             # base pose search / a robot on wheels: the next PLACEMENT (env.py:281-308)
            rejected.append(first if placed else info['ik_restarts'] - 1)
            first = first + 1 if placed else info['ik_restarts']
        info['rejected_restarts'] = rejected
        return st, info

    def _sample_from(self, seed, impairment_mode, gender_mode, max_restarts, first_restart, settled=None, fell=None):
        u = lambda idx: u01(seed, 0, idx)
        # bed bathing (FLAGS bit 4): the human lies where the rag doll of the second model came to rest (its record's q, float32)
        assert (settled is not None) == bool(self.xi('FLAGS') & 16)
        self.settled = None if settled is None else np.asarray(settled, dtype=np.float32)[:6 + self.xi('NJOINT') - 1].astype(np.float64)
        # arm manipulation (FLAGS bit 7; arm_manipulation.py:139-151): the arm's joint angles / velocities and the human's bodies after the fall, from the
        # fall model's record (this blob's layout); the fall model itself (bit 8) writes the record the arm falls from
        fall_stage = bool(self.xi('FLAGS') & 256)
        assert (fell is not None) == (bool(self.xi('FLAGS') & 128) and not fall_stage)
        self.fell = None if fell is None else np.asarray(fell, dtype=np.float32).astype(np.float64)
        friction = self.xf('FRIC_LO') + (self.xf('FRIC_HI') - self.xf('FRIC_LO')) * u(S_FRICTION)
        g = gender_mode if gender_mode >= 0 else (0 if u(S_GENDER) < 0.5 else 1)
        if impairment_mode >= 0:
            imp = impairment_mode
        else:
            nchoice = 4 if impairment_mode == MODE_RANDOM else 3
            imp = min(int(u(S_IMPAIRMENT) * nchoice), nchoice - 1)
        ls = 1.0 if imp != 1 else self.xf('LIMIT_LO') + (1.0 - self.xf('LIMIT_LO')) * u(S_LIMIT)
        strength = 1.0 if imp != 2 else self.xf('STRENGTH_LO') + (1.0 - self.xf('STRENGTH_LO')) * u(S_STRENGTH)
        tr = self.xf('TREMOR_RANGE')
        tremors = np.array([(2 * u(S_TREMOR + k) - 1) * tr if imp == 3 else 0.0 for k in range(self.nhdof)])
        hr = self.xf('HEAD_RANGE')
        head = [(2 * u(S_HEAD + k) - 1) * hr for k in range(3)]

        st = np.zeros(self.state_words, dtype=np.float32)
        si = st.view(np.int32)
        S = self.S
        bodies = [int(self.i[self.x0 + self.xi('OFF_BODIES') + k]) for k in range(self.nhuman)]
        dyn = [int(self.i[self.x0 + self.xi('OFF_DYN') + k]) for k in range(self.nhdof)]
        for k, link in enumerate(bodies):
            if self.fell is not None:                                      # the bodies did not move while the arm fell
                st[S['HUMAN'] + 7 * k:S['HUMAN'] + 7 * k + 7] = self.fell[S['HUMAN'] + 7 * k:S['HUMAN'] + 7 * k + 7]
                continue
            p, q = self.link_pose(g, link, ls, head)
            st[S['HUMAN'] + 7 * k:S['HUMAN'] + 7 * k + 3], st[S['HUMAN'] + 7 * k + 3:S['HUMAN'] + 7 * k + 7] = p, q
        target = np.zeros(3)
        if self.ti('HEAD_LINK') >= 0:                                      # tasks with a mouth target (feeding.py:184-196)
            hp, hq = self.link_pose(g, dyn[self.ti('HEAD_LINK') - self.nrobot], ls, head)
            target, _ = compose(hp, hq, self.tf('MOUTH_F' if g else 'MOUTH_M', 3), np.array([0, 0, 0, 1.0]))

        er = self.xf('EE_RANGE')
        target_ee = self.xf('EE_TARGET', 3) + np.array([(2 * u(S_EE + k) - 1) * er for k in range(3)])
        toc = self.xf('EE_QUAT', 4)
        two_arms, best2 = bool(self.xi('FLAGS') & 512), None               # a two-armed robot (arm_manipulation.py:159,165)
        target_ee2 = self.xf('EE_TARGET2', 3) + np.array([(2 * u(S_EE2 + k) - 1) * er for k in range(3)])
        self.arm = 0
        n_max = self.xi('IK_RESTARTS') if max_restarts is None else max_restarts
        thr = self.xf('IK_THRESH')
        best, best_d, ok, restarts = None, np.inf, False, 0
        base = (self.xf('BASE_POS', 3), self.xf('BASE_QUAT', 4))
        toc_info = None
        mobile, lift_dof, lift_q = bool(self.xi('FLAGS') & 8), -1, 0.0
        if fall_stage:                                                     # the robot is placed after the fall (arm_manipulation.py:162): parked, arm at mid range
            lower, upper = self.arm_limits()
            base = (self.xf('FALL_PARK', 3), np.array([0, 0, 0, 1.0]))
            best = np.where((lower > -1e9) & (upper < 1e9), 0.5 * (lower + upper), 0.0)
            if two_arms:
                self.arm = 1
                lower, upper = self.arm_limits()
                best2 = np.where((lower > -1e9) & (upper < 1e9), 0.5 * (lower + upper), 0.0)
                self.arm = 0
            ok, restarts, best_d, n_max = True, 0, 0.0, 0
        elif mobile:                                                       # a robot on wheels (env.py:282-293, stretch.py:58-62): no IK
            stream = T_STREAM0 + first_restart
            pr, yr = self.xf('TOC_POS_RANGE'), self.xf('TOC_YAW_RANGE')
            bp = self.xf('BASE_POS', 3) + np.array([(2 * u01(seed, stream, T_X) - 1) * pr, (2 * u01(seed, stream, T_Y) - 1) * pr, 0.0])
            base = (bp, q_axis_angle(np.array([0, 0, 1.0]), self.xf('TOC_YAW0') + (2 * u01(seed, stream, T_YAW) - 1) * yr))
            lift_dof, lift_q = self.xi('MOBILE_LIFT_DOF'), self.xf('MOBILE_LIFT') + (2 * u01(seed, stream, T_REST) - 1) * 0.1
            ok, restarts, best_d, n_max, best = True, 0, 0.0, 0, []
        elif self.xi('TOC_ATTEMPTS') > 0:                                  # a free-standing robot: base pose search instead of IK restarts
            if self.xi('TOC_GOAL_KIND') == 1:                              # feeding: the mouth (feeding.py:142)
                goals = [target]
            else:
                goals = [self.link_pose(g, int(self.i[self.x0 + (X_['TOC_GOAL_LINKS'] + k if k < 3 else X_['TOC_GOAL_LINK3'])]), ls, head)[0] + self.xf('TOC_GOAL_OFF', 3) for k in range(self.xi('TOC_NGOALS'))]
            gq = [self.f[self.x0 + X_['TOC_GOAL_QUAT'] + 4 * k:self.x0 + X_['TOC_GOAL_QUAT'] + 4 * k + 4].astype(np.float64) for k in range(3)] if self.xi('TOC_GOAL_ORIENT') else None
            must = 1
            if self.xi('TOC_GOAL_KIND') == 2:                              # drinking (drinking.py:143): the mouth (position) is a second START goal, the mouth with the
                goals, gq, must = [target, target], [None, toc], 3         # start pose's end-effector orientation the one further goal
            ok, base, best, restarts, ngoal, manip = self.toc(seed, first_restart, target_ee, toc, goals, gq, must, target_pos2=target_ee2 if two_arms else None)
            if two_arms:
                best, best2 = best
            best_d, n_max = float(ngoal), 0
            toc_info = dict(goals_reached=ngoal, manipulability=manip, base_pos=base[0], base_quat=base[1])
        for r in range(n_max):
            restarts = r + 1
            q, dpos, dor = self.restart(seed, r, target_ee, toc)
            met = dpos < thr and dor < thr                                 # robot.py:97
            if met and r < first_restart:                                  # collided there: `continue` (robot.py:110-112)
                continue
            if dpos < best_d:
                best, best_d = q, dpos
            if met:
                best, best_d, ok = q, dpos, True
                break
        nr, narm = self.nrobot, self.xi('NARM')
        qfull = np.zeros(self.ndof)
        ch = [] if mobile else self.chain()
        for d in range(nr):                                                # gripper (and joints outside the arm) opened instantly, feeding.py:143-144
            qfull[d] = min(max(self.rf(d, 'QT0'), self.rf(d, 'LOWER')), self.rf(d, 'UPPER'))
        for k, d in enumerate(ch):
            qfull[d] = best[k]
        if two_arms:
            self.arm = 1
            for k, d in enumerate(self.chain()):
                qfull[d] = best2[k]
            self.arm = 0
        if mobile:
            qfull[lift_dof] = lift_q
        for k, j in enumerate(dyn):
            qfull[nr + k] = self.joint_angle(g, j, ls, head)
        st[S['Q']:S['Q'] + self.ndof] = qfull
        st[S['QT']:S['QT'] + self.ndof] = qfull
        st[S['TREMOR']:S['TREMOR'] + self.nhdof] = tremors
        st[S['TREMOR'] + self.nhdof:S['TREMOR'] + 2 * self.nhdof] = qfull[nr:]
        if self.fell is not None:                                          # the arm keeps its velocity, its hold the pose it was given before the fall
            for key in ('QD', 'QT'):
                st[S[key] + nr:S[key] + self.ndof] = self.fell[S[key] + nr:S[key] + self.ndof]
            st[S['TREMOR'] + self.nhdof:S['TREMOR'] + 2 * self.nhdof] = self.fell[S['TREMOR'] + self.nhdof:S['TREMOR'] + 2 * self.nhdof]
        st[S['BASE']:S['BASE'] + 3], st[S['BASE'] + 3:S['BASE'] + 7] = base
        # tool in the hand (tool.py:49-62)
        pe, oe = self.mobile_fk(lift_dof, lift_q, base) if mobile else self.arm_fk(best, base)[:2]
        tp, tq = compose(pe, oe, self.tf('TOOL_POS', 3), self.tf('TOOL_QUAT', 4))
        fr = lambda b: S['FREE'] + 13 * b
        for b in range(self.nfree):
            st[fr(b) + 6] = 1.0
        if self.nfree == 0:                                                # dressing: no tool, no free bodies
            self.tool_body = -1
        fo = int(self.i[H_OFF_FREE]) + max(self.tool_body, 0) * F['STRIDE']
        refp, refq = self.f[fo + F['REFPOS']:fo + F['REFPOS'] + 3].astype(np.float64), self.f[fo + F['REFQUAT']:fo + F['REFQUAT'] + 4].astype(np.float64)
        cp, cq = tp, tq
        if np.any(refp != 0) or refq[3] != 1:                              # a welded tool (scratcher): the record holds the COM frame
            qi = np.array([-refq[0], -refq[1], -refq[2], refq[3]])
            cp, cq = compose(tp, tq, -qrot(qi, refp), qi)
        if self.tool_body >= 0:
            st[fr(self.tool_body):fr(self.tool_body) + 3], st[fr(self.tool_body) + 3:fr(self.tool_body) + 7] = cp, cq
        if two_arms:                                                       # tool_left in the left hand (arm_manipulation.py:156)
            self.arm = 1
            pe2, oe2 = self.arm_fk(best2, base)[:2]
            self.arm = 0
            t2 = self.ti('TOOL2_BODY')
            tp2, tq2 = compose(pe2, oe2, self.tf('TOOL2_POS', 3), self.tf('TOOL2_QUAT', 4))
            fo2 = int(self.i[H_OFF_FREE]) + t2 * F['STRIDE']
            refp2, refq2 = self.f[fo2 + F['REFPOS']:fo2 + F['REFPOS'] + 3].astype(np.float64), self.f[fo2 + F['REFQUAT']:fo2 + F['REFQUAT'] + 4].astype(np.float64)
            if np.any(refp2 != 0) or refq2[3] != 1:
                qi2 = np.array([-refq2[0], -refq2[1], -refq2[2], refq2[3]])
                tp2, tq2 = compose(tp2, tq2, -qrot(qi2, refp2), qi2)
            st[fr(t2):fr(t2) + 3], st[fr(t2) + 3:fr(t2) + 7] = tp2, tq2
        # bowl (furniture.py:32-34): base frame -> COM frame
        bb = self.xi('BOWL_BODY')
        if bb >= 0:
            br = self.xf('BOWL_RANGE')
            bowl = self.xf('BOWL_POS', 3) + np.array([(2 * u(S_BOWL) - 1) * br, (2 * u(S_BOWL + 1) - 1) * br, 0.0])
            fo = int(self.i[H_OFF_FREE]) + bb * F['STRIDE']
            refp, refq = self.f[fo + F['REFPOS']:fo + F['REFPOS'] + 3], self.f[fo + F['REFQUAT']:fo + F['REFQUAT'] + 4]
            qi = np.array([-refq[0], -refq[1], -refq[2], refq[3]])
            cp, cq = compose(bowl, np.array([0, 0, 0, 1.0]), -qrot(qi, refp), qi)
            st[fr(bb):fr(bb) + 3], st[fr(bb) + 3:fr(bb) + 7] = cp, cq
        # food grid above the spoon (feeding.py:158-166)
        rf_, fo3 = self.xf('FOOD_R'), self.xf('FOOD_OFF', 3)
        k = 0
        for a in range(2):
            for b in range(2):
                for c in range(2):
                    if k < self.nfood:
                        st[fr(self.food0 + k):fr(self.food0 + k) + 3] = np.array([a * 2 * rf_, b * 2 * rf_, c * 2 * rf_]) + fo3 + tp
                    k += 1
        e = S['ENV']
        st[e + 0] = friction
        si[e + 1] = g
        st[e + 2:e + 5] = target
        si[e + 5] = si[e + 6] = (1 << self.nfood) - 1
        si[e + 7] = si[e + 8] = 0
        si[e + 9] = (seed * 2654435761 + 12345) & 0x7FFFFFFF
        si[e + 10] = (seed ^ 0x5bd1e995) & 0x7FFFFFFF
        xflags = self.xi('FLAGS')
        si[e + 11] = 1 if xflags & (6 | 128) else self.nfood
        if xflags & 64:                                                    # bed bathing: every wiping target alive (bed_bathing.py:173-188)
            to = int(self.i[H_OFF_TASK])
            nt = int(self.i[to + 52 + 2 * g]) + int(self.i[to + 52 + 2 * g + 1])      # AGX_T_NT
            si[e + 11] = nt
            ts = int(self.i[H_S_TASK])
            for t in range(6):
                st.view(np.uint32)[ts + t] = 0xffffffff if nt >= 32 * (t + 1) else ((1 << (nt - 32 * t)) - 1 if nt > 32 * t else 0)
        if int(self.i[H_TASK_KIND]) == 5:                                   # drinking: every water particle in self.waters / self.waters_active (drinking.py:168-172)
            oc = int(self.i[H_OFF_CLOTH])
            nw = int(self.i[oc + 0])                                       # AGX_CL_NN
            si[e + 11] = nw
            ts = int(self.i[H_S_TASK])
            for t in range(2):
                m = 0xffffffff if nw >= 32 * (t + 1) else ((1 << (nw - 32 * t)) - 1 if nw > 32 * t else 0)
                st.view(np.uint32)[ts + 0 + t] = st.view(np.uint32)[ts + 2 + t] = m      # AGX_DK_ALIVE, AGX_DK_ACTIVE
        coop = self.ti('COOP') == 1
        agent = imp == 3 or coop
        si[e + 12] = 0 if (agent or xflags & 1) else (((1 << self.nhdof) - 1) << nr)     # human.py:104-110
        st[e + 13] = ls
        if not agent and self.xf('REACTIVE_KP') > 0:                       # the reactive hold of setup_joints (human.py:124-127)
            st[e + 14], st[e + 15] = self.xf('REACTIVE_KP'), self.xf('REACTIVE_MAXF') * strength
        limb, target_on_arm = None, None
        if xflags & 2:                                                     # generate_target (scratch_itch.py:134-146, util.py:58-78)
            limb = 0 if u(S_LIMB) < 0.5 else 1
            dims = self.tf('SI_LIMB_DIMS', 8).astype(np.float64)
            length, radius = dims[4 * g + 2 * limb], dims[4 * g + 2 * limb + 1]
            rl = radius + (length - radius) * u(S_TARGET_LEN)
            th = 2 * np.pi * u(S_TARGET_TH)
            axis, ortho, normal = np.array([0, 0, -1.0]), np.array([0, -1.0, 0]), np.array([-1.0, 0, 0])
            target_on_arm = rl * axis + radius * np.cos(th) * ortho + radius * np.sin(th) * normal
            ts = int(self.i[H_S_TASK])
            st[ts:ts + 3] = target_on_arm
            si[ts + 3] = limb
        if xflags & 4:                                                     # dressing: settle gravity, garment offset (dressing.py:146-149,178)
            ts = int(self.i[H_S_TASK])
            st[ts + 0] = self.xf('CLOTH_GRAVITY_SETTLE')
            st[ts + 3:ts + 6] = pe - self.xf('CLOTH_ORIG_POS', 3)
        return st, dict(gender=g, impairment=imp, limit_scale=ls, strength=strength, tremors=tremors, ik_ok=ok,
                        ik_restarts=restarts, ik_pos_err=best_d, target_ee=target_ee, head=head, limb=limb, target_on_arm=target_on_arm, toc=toc_info)
