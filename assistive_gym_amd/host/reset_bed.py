"""Host-side reset for BedBathingSawyer-v1: produces post-reset state records (the stepper's input).

Follows the order of BedBathingEnv.reset (assistive_gym/envs/bed_bathing.py:112-171) and what it calls:
build_assistive_env('bed', fixed_human_base=False) (envs/env.py:114-134: plane friction, Human.init draws,
agents/human.py:72-102), the human posed in the air and its joints perturbed by U(-0.1, 0.1) (bed_bathing.py:119-127),
the settle onto the bed (bed_bathing.py:129-137), the target end-effector pose (bed_bathing.py:147-148),
init_robot_pose -> Robot.position_robot_toc (env.py:276-310, robot.py:123-215: 50 random base poses scored by goals
reached and joint-limit-weighted kinematic isotropy), the gripper (bed_bathing.py:156), generate_targets
(bed_bathing.py:173-188) and the per-body gravities (bed_bathing.py:160-165, compiled into the blob).

Reset is outside the kernel scope; only its RESULT feeds the stepper.  What is NOT the reference's:
  * Bullet's IK is replaced by damped least squares (as in host/kin.py); the TOC search keeps the reference's structure
    (50 attempts, start pose + 3 position-only goals, success threshold 0.03, JLWKI score);
  * the 100-step rag-doll settle of the 47-DoF floating human (bed_bathing.py:129-131) is produced by `settle`:
    'ragdoll' runs it: the bed_settle model (compiler.compile_bed_settle: the human as one floating articulated body, its own
    kernel variant) is stepped 100 times by a `settler` (RagdollSettler: the device, through agx_settle) and the resting base
    pose and joint angles are read back; 'drop' (the default without a device) lowers the posed human rigidly until its first
    collider touches the mattress -- a kinematic stand-in that keeps the perturbed joint angles but lets limbs float above
    the bed by the difference of their radii;
  * the collision rejection loop of init_robot_pose (env.py:299-308) is not run.
"""
import numpy as np

from ..model import compiler as L
from ..model import xform as X
from ..model.human import HumanModel

D = np.deg2rad


class ArmChain:
    """Batched (numpy) kinematics of the serial chain base -> end-effector link of the compiled robot."""

    def __init__(self, blob, second=False):
        """second: the chain to the second end effector (AGX_T_EE2_*: the left arm of a two-armed robot holding tool_left)"""
        self.blob = blob
        ee = blob.task_i('EE2_LINK' if second else 'EE_LINK')
        chain = []
        d = ee
        while d >= 0:
            chain.append(d)
            d = blob.robot_i(d, 'PARENT')
        self.chain = chain[::-1]                                  # root -> ee
        assert all(blob.robot_i(d, 'JTYPE') == 0 for d in self.chain)
        self.tpos = np.array([blob.robot_f(d, 'TPOS', 3) for d in self.chain])
        self.tR = np.array([X.quat_to_mat(blob.robot_f(d, 'TQUAT', 4)) for d in self.chain])
        self.axis = np.array([blob.robot_f(d, 'AXIS', 3) for d in self.chain])
        self.lower = np.array([blob.robot_f(d, 'LOWER') for d in self.chain])
        self.upper = np.array([blob.robot_f(d, 'UPPER') for d in self.chain])
        self.act = [blob.robot_i(d, 'ACT') for d in self.chain]
        assert all(a >= 0 for a in self.act), 'every joint between the base and the end effector is an arm joint'
        self.ee_pos = blob.task_f('EE2_POS' if second else 'EE_POS', 3)
        self.ee_R = X.quat_to_mat(blob.task_f('EE2_QUAT' if second else 'EE_QUAT', 4))
        self.n = len(self.chain)

    @staticmethod
    def _rot(axis, ang):
        """Rodrigues, batched over ang (B,) for one axis (3,) -> (B, 3, 3)"""
        a = axis / np.linalg.norm(axis)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        s, c = np.sin(ang)[:, None, None], np.cos(ang)[:, None, None]
        return np.eye(3)[None] + s * K[None] + (1 - c) * (K @ K)[None]

    def fk(self, base_pos, base_R, q):
        """base_pos (B,3), base_R (B,3,3), q (B,n) -> ee pos (B,3), ee R (B,3,3), joint origins (B,n,3), joint axes in world (B,n,3)"""
        B = q.shape[0]
        p, R = base_pos.copy(), base_R.copy()
        orig, axw = np.zeros((B, self.n, 3)), np.zeros((B, self.n, 3))
        for k in range(self.n):
            p = p + np.einsum('bij,j->bi', R, self.tpos[k])
            R = R @ self.tR[k][None]
            orig[:, k] = p
            axw[:, k] = np.einsum('bij,j->bi', R, self.axis[k])
            R = R @ self._rot(self.axis[k], q[:, k])
        pe = p + np.einsum('bij,j->bi', R, self.ee_pos)
        return pe, R @ self.ee_R[None], orig, axw

    def jacobian(self, pe, orig, axw):
        Jl = np.cross(axw, pe[:, None, :] - orig)                 # (B,n,3)
        return np.concatenate([Jl.transpose(0, 2, 1), axw.transpose(0, 2, 1)], axis=1)    # (B,6,n)

    def ik(self, base_pos, base_R, q0, target_pos, target_R=None, iters=100, damping=0.05, maxstep=0.5):
        """Damped least squares, batched; target_R None = position only.  Returns q (B,n)."""
        q = q0.copy()
        for _ in range(iters):
            pe, Re, orig, axw = self.fk(base_pos, base_R, q)
            J = self.jacobian(pe, orig, axw)
            e = target_pos - pe
            if target_R is not None:
                Rerr = target_R @ Re.transpose(0, 2, 1)
                w = 0.5 * np.stack([Rerr[:, 2, 1] - Rerr[:, 1, 2], Rerr[:, 0, 2] - Rerr[:, 2, 0], Rerr[:, 1, 0] - Rerr[:, 0, 1]], axis=1)
                e = np.concatenate([e, w], axis=1)
            else:
                J = J[:, :3]
            A = J @ J.transpose(0, 2, 1) + damping ** 2 * np.eye(J.shape[1])[None]
            dq = np.einsum('bji,bj->bi', J, np.linalg.solve(A, e[..., None])[..., 0])
            step = np.abs(dq).max(axis=1, keepdims=True)
            dq = np.where(step > maxstep, dq * (maxstep / np.maximum(step, 1e-30)), dq)
            q = np.clip(q + dq, self.lower[None], self.upper[None])
        return q


def mat_to_quat_batch(R):
    return np.array([X.mat_to_quat(r) for r in R])


def joint_limited_weighting(q, lower, upper):
    """Robot.joint_limited_weighting (robot.py:217-228), batched over q (B,n) -> diagonal weights (B,n)"""
    phi, lam = 0.5, 0.05
    qr = 0.5 * (upper - lower)
    w = 1.0 - np.power(phi, (qr - np.abs(qr - q + lower)) / (lam * qr) + 1)
    return np.maximum(w, 0.001)


def settle_record(sblob, row, gender, limit_scale, base_pos, base_rpy, hq, plane_friction):
    """state record of the bed_settle model: the posed human in the air (bed_bathing.py:121,127), at rest"""
    v = sblob.view(row)
    row[:] = 0
    joints = sblob.meta['settle_joints']
    q = np.concatenate([base_pos, [base_rpy[2], base_rpy[1], base_rpy[0]], [hq[j] for j in joints]])
    v['q'][0] = q
    v['qt'][0] = q
    v['human'][0, 0, 3:] = [0, 0, 0, 1]                       # the world anchor of the virtual root joints
    v['base'][0, 3:] = [0, 0, 0, 1]
    v['limit_scale'][0] = limit_scale
    v['plane_friction'][0] = plane_friction
    v['gender'][0] = 0 if gender == 'male' else 1
    return row


def settled_pose(sblob, row, hm):
    """(base_pos, base_quat, hq[hm.n]) of a settle record"""
    v = sblob.view(row)
    q = v['q'][0].astype(np.float64)
    hq = np.zeros(hm.n)
    for k, j in enumerate(sblob.meta['settle_joints']):
        hq[j] = q[6 + k]
    return q[:3].copy(), X.quat_from_rpy([q[5], q[4], q[3]]), hq


class RagdollSettler:
    """Runs the rag-doll settle on the device: an agx context on the bed_settle model, agx_settle for 100 simulation steps
    (bed_bathing.py:130-131).  Fails loudly without a GPU (the product path has no CPU fallback)."""

    def __init__(self, n_envs, device=0, steps=100):
        from ..blob import ModelBlob
        from ..libagx import Stepper
        self.blob = ModelBlob.load('bed_settle')
        self.ctx = Stepper(self.blob, n_envs, device)
        self.n, self.steps = n_envs, steps

    def __call__(self, states):
        n = len(states)
        assert n <= self.n
        buf = self.blob.new_state(self.n)
        buf[:n] = states
        buf[n:] = states[:1]
        self.ctx.set_state(buf)
        self.ctx.settle(self.steps)
        self.ctx.L.agx_synchronize(self.ctx.h, None)
        return self.ctx.get_state()[:n]


class DeviceCollisionChecker:
    """states -> AGX_COLLIDE_* flags per state, from the stepper's own collision pass on the device (agx_check_collisions): the
    collision rejection of init_robot_pose / ik_random_restarts (env.py:299-308, robot.py:103-108) for resets sampled on the host.
    Fails loudly without a GPU."""

    def __init__(self, blob, n_envs, device=0):
        from ..libagx import Stepper
        self.blob, self.n = blob, n_envs
        self.ctx = Stepper(blob, n_envs, device)

    def __call__(self, states):
        out = np.zeros(len(states), dtype=np.uint8)
        for i0 in range(0, len(states), self.n):
            chunk = states[i0:i0 + self.n]
            buf = self.blob.new_state(self.n)
            buf[:len(chunk)] = chunk
            buf[len(chunk):] = chunk[:1]
            self.ctx.set_state(buf)
            out[i0:i0 + len(chunk)] = self.ctx.check_collisions()[:len(chunk)]
        return out


def placement_rng(rng, env_seed, attempt):
    """the random stream of the robot's placement: the sampler's own stream on the first attempt, a stream of its own on the re-draws
    of init_robot_pose's rejection loop (env.py:281), so that nothing else of the episode changes"""
    return rng if attempt == 0 else np.random.RandomState((env_seed * 7919 + 104729 * attempt) & 0x7FFFFFFF)


def reject_collisions(states, flags_of, resample, max_iterations=3):
    """init_robot_pose's loop (env.py:281-308): states whose robot or tool touches the human / the furniture (or whose arm is folded into
    itself) are placed again, at most max_iterations placements in total; resample(i, attempt) rewrites states[i].  Returns the final
    flags (a state still colliding after the last attempt is kept, as in the reference)."""
    flags = flags_of(states)
    for attempt in range(1, max_iterations):
        bad = np.flatnonzero(flags)
        if not len(bad):
            break
        for i in bad:
            resample(int(i), attempt)
        flags[bad] = flags_of(states[bad])
    return flags


class BedBathingSawyerReset:
    def __init__(self, blob, settle='drop'):
        assert blob.task_kind == L.TASK_BED_BATHING
        self.blob = blob
        m = blob.meta
        self.mount = m.get('mount', 'toc')
        if self.mount == 'mobile':                                  # the Stretch: no arm chain to solve (env.py:282-293)
            from .reset import MobilePlacement
            self.arm, self.mobile = None, MobilePlacement(blob)
        else:
            self.arm = ArmChain(blob)
        self.human_bodies = blob.meta['human_bodies']
        self.human_dyn = blob.meta['human_dynamic_joints']
        self.settle = settle
        self.toc_base = np.array([-0.85, -0.4, 0]) + np.array(m.get('toc_base', [-0.2, 0, 0.975]))      # robot.py:142 + toc_base_pos_offset (sawyer.py:37)
        self.ee_R = X.quat_to_mat(X.quat_from_rpy(m.get('ee_rpy', [0, np.pi / 2.0, 0])))                # toc_ee_orient_rpy (sawyer.py:43)
        r = blob.meta['ranges']
        self._bed = [blob.collider(c)['verts'] for c in range(*r['bed'])]
        self._bed_box = np.array([[v.min(0), v.max(0)] for v in self._bed])
        self._hm = {}

    def _human(self, gender, limit_scale):
        key = (gender, round(float(limit_scale), 9))
        if key not in self._hm:
            if len(self._hm) > 64:
                self._hm.clear()
            self._hm[key] = HumanModel(gender, limit_scale)
        return self._hm[key]

    # ---- the lying human --------------------------------------------------------------------------------
    def _bed_top(self, x, y):
        """height of the bed's collision geometry under (x, y): top of the hulls whose footprint box contains the point"""
        b = self._bed_box
        inside = (b[:, 0, 0] <= x) & (x <= b[:, 1, 0]) & (b[:, 0, 1] <= y) & (y <= b[:, 1, 1]) & (b[:, 1, 2] < 0.9)   # not the head / foot boards
        return float(b[inside, 1, 2].max()) + L.HULL_MARGIN if inside.any() else 0.0

    def _drop(self, hm, base_pos, base_quat, hq):
        """lowers the rigidly posed human until its first collider rests on the bed"""
        pos, quat = hm.fk(base_pos, base_quat, hq)
        dz = -np.inf
        for link, kind, data in hm.colliders():
            lp, lq = (base_pos, base_quat) if link < 0 else (pos[link], quat[link])
            if kind == 'capsule':
                pts, r = X.apply(lp, lq, np.stack([data[0], data[1]])), data[2]
            elif kind == 'sphere':
                pts, r = X.apply(lp, lq, data[0][None]), data[1]
            else:
                continue            # the head mesh: it rests on the pillow region, the neck / chest decide
            for p in pts:
                dz = max(dz, self._bed_top(p[0], p[1]) + r - p[2])
        return base_pos + np.array([0, 0, dz])

    # ---- TOC base pose search (robot.py:123-215) ---------------------------------------------------------
    def _toc(self, rng, start_pos, goals, attempts=50, goal_Rs=None, right_side=True):
        """goal_Rs: optional end-effector orientation (3x3) per goal, None = position only; right_side=False: the base is drawn on the
        human's left and turned by pi (env.py:298 base_euler_orient, robot.py:143)"""
        arm = self.arm
        A = attempts
        rp = np.stack([rng.uniform(-0.5, 0, size=A) if right_side else rng.uniform(0, 0.5, size=A), rng.uniform(-0.5, 0.5, size=A), np.zeros(A)], axis=1)     # random_position 0.5
        yaw = (0.0 if right_side else np.pi) + D(rng.uniform(-30, 30, size=A))                              # random_rotation 30
        # the reference draws position and yaw alternately per attempt; the draws here are made in two blocks (not stream compatible anyway)
        base_pos = self.toc_base[None] + rp
        base_R = np.array([X.quat_to_mat(X.quat_from_rpy([0, 0, y])) for y in yaw])
        ng = 1 + len(goals)
        lo = np.where(arm.lower < -1e9, -2 * np.pi, arm.lower)
        hi = np.where(arm.upper > 1e9, 2 * np.pi, arm.upper)
        q0 = rng.uniform(lo, hi, size=(A, ng, arm.n))                                                         # agent.py:263 rest poses
        reached = np.zeros((A, ng), dtype=bool)
        jl = np.zeros((A, ng))
        qsol = np.zeros((A, ng, arm.n))
        for g in range(ng):
            tp = np.repeat((start_pos if g == 0 else goals[g - 1])[None], A, axis=0)
            gR = self.ee_R if g == 0 else (goal_Rs[g - 1] if goal_Rs is not None else None)
            tR = np.repeat(gR[None], A, axis=0) if gR is not None else None
            q = arm.ik(base_pos, base_R, q0[:, g], tp, tR, iters=100)                                        # max_ik_iterations=100
            pe, Re, orig, axw = arm.fk(base_pos, base_R, q)
            ok = np.linalg.norm(tp - pe, axis=1) < 0.03                                                      # robot.py:97 success_threshold
            if tR is not None:
                qe, qt = mat_to_quat_batch(Re), X.mat_to_quat(gR)
                dq = np.minimum(np.linalg.norm(qe - qt[None], axis=1), np.linalg.norm(qe + qt[None], axis=1))
                ok &= dq < 0.03
            J = arm.jacobian(pe, orig, axw)
            W = joint_limited_weighting(q, arm.lower, arm.upper)
            M = np.einsum('bij,bj,bkj->bik', J, W, J)
            det = np.maximum(np.linalg.det(M), 0)
            jl[:, g] = np.power(det, 1.0 / 6) / (np.trace(M, axis1=1, axis2=2) / 6)                          # robot.py:189-191
            if g == 0 and getattr(self, 'self_guard', False):
                ok &= ~self._arm_in_pedestal(base_pos, base_R, orig, pe)
            reached[:, g] = ok
            qsol[:, g] = q
        valid = reached[:, 0]                                   # the start goal must be reachable (robot.py:196-200)
        ngoal = np.where(valid, reached.sum(1), -1)
        manip = np.where(valid, (jl * reached).sum(1), -np.inf)
        best = max(range(A), key=lambda a: (ngoal[a], manip[a]))   # first of equals = the earliest attempt, as the strict > of robot.py:204
        if ngoal[best] <= 0:
            return None
        return base_pos[best], X.mat_to_quat(base_R[best]), qsol[best, 0], int(ngoal[best]), float(manip[best])

    def _arm_in_pedestal(self, base_pos, base_R, orig, pe, margin=0.09):
        """start poses whose arm folds into the robot's own pedestal (boxes of the base colliders grown by a link radius): Bullet's
        null-space IK with rest poses does not produce those elbow-down solutions, the damped least squares here can"""
        if not hasattr(self, '_ped'):
            r = self.blob.meta['ranges']['robot_base']
            self._ped = np.array([[self.blob.collider(c)['verts'].min(0) - margin, self.blob.collider(c)['verts'].max(0) + margin] for c in range(*r)])
        pts = np.concatenate([orig[:, 2:], 0.5 * (orig[:, 2:-1] + orig[:, 3:]), pe[:, None]], axis=1)          # joint origins past the shoulder, midpoints, ee
        loc = np.einsum('bji,bkj->bki', base_R, pts - base_pos[:, None])
        inside = (loc[:, :, None, :] >= self._ped[None, None, :, 0]) & (loc[:, :, None, :] <= self._ped[None, None, :, 1])
        return inside.all(-1).any(-1).any(-1)

    # ---- wheelchair-mounted robots: IK with random restarts from a fixed base (robot.py:84-121) ------------------------------------
    def _near_human(self, hm, hpos, hquat, hbase, pts, margin=0.07):
        """pts (B, K, 3): is any point closer than `margin` (an arm link's radius + clearance) to a capsule / sphere of the human?  The
        stand-in for `get_closest_points(human, distance=0)` inside ik_random_restarts (robot.py:103-108) on the host."""
        hit = np.zeros(pts.shape[0], dtype=bool)
        for link, kind, data in hm.colliders():
            lp, lq = (hbase, np.array([0, 0, 0, 1.0])) if link < 0 else (hpos[link], hquat[link])
            if kind == 'capsule':
                a, b = X.apply(lp, lq, np.stack([data[0], data[1]]))
                r = data[2]
            elif kind == 'sphere':
                a = b = X.apply(lp, lq, data[0][None])[0]
                r = data[1]
            else:
                continue
            ab = b - a
            t = np.clip(((pts - a) @ ab) / max(float(ab @ ab), 1e-12), 0, 1)
            d = np.linalg.norm(pts - (a + t[..., None] * ab), axis=-1) - r
            hit |= (d < margin).any(axis=1)
        return hit

    def _mounted_ik(self, rng, target_pos, base_pos, base_quat, human=None, restarts=64, rounds=4):
        """Robot.ik_random_restarts (robot.py:84-121) from the fixed base: random rest poses until the end effector is within 0.01 of the
        target pose and the arm is clear of the human; `restarts` of them are solved at once, the first that qualifies is taken; when
        none does, the closest one (robot.py:117-121)"""
        arm = self.arm
        bp = np.repeat(np.asarray(base_pos, dtype=np.float64)[None], restarts, axis=0)
        bR = np.repeat(X.quat_to_mat(base_quat)[None], restarts, axis=0)
        lo = np.where(arm.lower < -1e9, -2 * np.pi, arm.lower)
        hi = np.where(arm.upper > 1e9, 2 * np.pi, arm.upper)
        tp, tR = np.repeat(target_pos[None], restarts, axis=0), np.repeat(self.ee_R[None], restarts, axis=0)
        qt = X.mat_to_quat(self.ee_R)
        best = None
        for _ in range(rounds):
            q = arm.ik(bp, bR, rng.uniform(lo, hi, size=(restarts, arm.n)), tp, tR, iters=200)
            pe, Re, orig, _ = arm.fk(bp, bR, q)
            qe = mat_to_quat_batch(Re)
            err = np.linalg.norm(tp - pe, axis=1) + np.minimum(np.linalg.norm(qe - qt[None], axis=1), np.linalg.norm(qe + qt[None], axis=1))
            ok = (np.linalg.norm(tp - pe, axis=1) < 0.01) & (np.minimum(np.linalg.norm(qe - qt[None], axis=1), np.linalg.norm(qe + qt[None], axis=1)) < 0.01)
            if human is not None and ok.any():
                pts = np.concatenate([orig[:, 1:], 0.5 * (orig[:, 1:-1] + orig[:, 2:]), pe[:, None], 0.5 * (orig[:, -1:] + pe[:, None])], axis=1)
                ok &= ~self._near_human(*human, pts)
            k = int(np.argmax(ok)) if ok.any() else int(np.argmin(err))
            if best is None or err[k] < best[0]:
                best = (float(err[k]), q[k].copy(), bool(ok[k]))
            if ok.any():
                break
        return np.asarray(base_pos, dtype=np.float64).copy(), np.asarray(base_quat, dtype=np.float64), best[1], 1 if best[2] else 0, 0.0

    def _toc_dual(self, rng, arms, starts, goals, attempts=50):
        """Robot.position_robot_toc with arms = ['right', 'left'] (robot.py:123-215): one base pose for both arms; the goals an arm reaches
        and its manipulability add up over the arms, both start poses must be reachable.  arms: [ArmChain, ArmChain]; starts[i]: start
        position of arm i (orientation self.ee_R); goals[i]: position-only goals of arm i.  Returns (base_pos, base_quat, [q_arm_i], goals
        reached, manipulability) or None."""
        A = attempts
        rp = np.stack([rng.uniform(-0.5, 0, size=A), rng.uniform(-0.5, 0.5, size=A), np.zeros(A)], axis=1)
        yaw = D(rng.uniform(-30, 30, size=A))
        base_pos = self.toc_base[None] + rp
        base_R = np.array([X.quat_to_mat(X.quat_from_rpy([0, 0, y])) for y in yaw])
        ngoal, manip, valid = np.zeros(A), np.zeros(A), np.ones(A, dtype=bool)
        qstart = []
        for arm, start, gl in zip(arms, starts, goals):
            lo = np.where(arm.lower < -1e9, -2 * np.pi, arm.lower)
            hi = np.where(arm.upper > 1e9, 2 * np.pi, arm.upper)
            ng = 1 + len(gl)
            q0 = rng.uniform(lo, hi, size=(A, ng, arm.n))
            for g in range(ng):
                tp = np.repeat((start if g == 0 else gl[g - 1])[None], A, axis=0)
                tR = np.repeat(self.ee_R[None], A, axis=0) if g == 0 else None
                q = arm.ik(base_pos, base_R, q0[:, g], tp, tR, iters=100)
                pe, Re, orig, axw = arm.fk(base_pos, base_R, q)
                ok = np.linalg.norm(tp - pe, axis=1) < 0.03
                if tR is not None:
                    qe, qt = mat_to_quat_batch(Re), X.mat_to_quat(self.ee_R)
                    ok &= np.minimum(np.linalg.norm(qe - qt[None], axis=1), np.linalg.norm(qe + qt[None], axis=1)) < 0.03
                J = arm.jacobian(pe, orig, axw)
                W = joint_limited_weighting(q, arm.lower, arm.upper)
                M = np.einsum('bij,bj,bkj->bik', J, W, J)
                det = np.maximum(np.linalg.det(M), 0)
                jl = np.power(det, 1.0 / 6) / (np.trace(M, axis1=1, axis2=2) / 6)
                ngoal += ok
                manip += np.where(ok, jl, 0.0)
                if g == 0:
                    valid &= ok
                    qstart.append(q)
        ngoal = np.where(valid, ngoal, -1)
        manip = np.where(valid, manip, -np.inf)
        best = max(range(A), key=lambda a: (ngoal[a], manip[a]))
        if ngoal[best] <= 0:
            return None
        return base_pos[best], X.mat_to_quat(base_R[best]), [q[best] for q in qstart], int(ngoal[best]), float(manip[best])

    def sample(self, rng, state_row, env_seed=0, impairment='random', gender='random', info=None, human_q_override=None):
        """Fill one state record (float32 view of length state_words) in place, with the 'drop' stand-in for the settle."""
        pre = self.pre_settle(rng, impairment, gender, human_q_override)
        pre['base_pos'] = self._drop(pre['hm'], pre['base_pos'], pre['base_quat'], pre['hq'])
        return self.post_settle(rng, state_row, pre, env_seed, info)

    def pre_settle(self, rng, impairment='random', gender='random', human_q_override=None):
        """the draws of build_assistive_env and the posed human in the air (bed_bathing.py:114-127)"""
        nh = self.blob.nhdof
        plane_friction = rng.uniform(0.025, 0.5)                                   # env.py:120
        if gender not in ('male', 'female'):
            gender = rng.choice(['male', 'female'])                                # human.py:76-77
        if impairment == 'random':
            impairment = rng.choice(['none', 'limits', 'weakness', 'tremor'])      # human.py:80-81
        elif impairment == 'no_tremor':
            impairment = rng.choice(['none', 'limits', 'weakness'])
        limit_scale = 1.0 if impairment != 'limits' else rng.uniform(0.5, 1.0)     # human.py:85
        strength = 1.0 if impairment != 'weakness' else rng.uniform(0.25, 1.0)     # human.py:86
        tremors = np.zeros(nh)
        if impairment == 'tremor':
            tremors = rng.uniform(D(-10), D(10), size=nh)                          # human.py:91-92 (the head is not controllable)
        rng.uniform(0.4, 0.8)                                                      # skin colour, human_creation.py:63
        hm = self._human(gender, limit_scale)
        # bed_bathing.py:119-127: pose in the air, every motor joint ~ U(-0.1, 0.1) (this overwrites the 30 degree shoulder preset), limits enforced
        hq = np.zeros(hm.n)
        movable = [j for j in range(hm.n) if hm.jtype[j] == 'r']
        hq[movable] = rng.uniform(-0.1, 0.1, size=len(movable))
        for j, a in (human_q_override or {}).items():          # tests only: e.g. an abducted arm
            hq[j] = a
        hq = hm.clamp(hq)
        base_rpy = np.array([-np.pi / 2.0, 0, 0])
        return dict(plane_friction=plane_friction, gender=gender, impairment=impairment, limit_scale=limit_scale, strength=strength,
                    tremors=tremors, hm=hm, hq=hq, base_pos=np.array([-0.15, 0.2, 0.95]), base_rpy=base_rpy, base_quat=X.quat_from_rpy(base_rpy))

    def post_settle(self, rng, state_row, pre, env_seed=0, info=None, attempt=0):
        """everything after the settle (bed_bathing.py:133-171), from the resting pose in `pre`"""
        b = self.blob
        v = b.view(state_row)
        nr, nh = b.nrobot, b.nhdof
        plane_friction, gender, impairment, limit_scale, strength, tremors = (pre[k] for k in ('plane_friction', 'gender', 'impairment', 'limit_scale', 'strength', 'tremors'))
        hm, hq, base_pos, base_quat = pre['hm'], pre['hq'], pre['base_pos'], pre['base_quat']
        hpos, hquat = hm.fk(base_pos, base_quat, hq)
        for k, link in enumerate(self.human_bodies):
            if link < 0:
                v['human'][0, k, :3], v['human'][0, k, 3:] = base_pos, base_quat
            else:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hpos[link], hquat[link]
        shoulder, elbow, wrist = hpos[5], hpos[7], hpos[9]                         # bed_bathing.py:139-141
        if attempt == 0:
            pre['target_ee_pos'] = np.array([-0.6, 0.2, 1]) + rng.uniform(-0.05, 0.05, size=3)    # bed_bathing.py:147
        target_ee_pos = pre['target_ee_pos']
        prng = placement_rng(rng, env_seed, attempt)
        toc = self.mobile.draw(prng) + (0, 0.0) if self.mount == 'mobile' else None
        for _ in range(4):
            if toc is not None:
                break
            toc = self._toc(prng, target_ee_pos, [shoulder, elbow, wrist])
        assert toc is not None, 'no reachable base pose found'
        rb_pos, rb_quat, q_arm, ngoal, manip = toc
        if self.mount == 'mobile':
            q = q_arm.copy()
        else:
            q = np.zeros(nr)
            for k, d in enumerate(self.arm.chain):
                q[d] = q_arm[k]
        for d in range(nr):                                                        # gripper open position, set instantly (bed_bathing.py:156)
            if b.robot_i(d, 'ACT') < 0:
                q[d] = min(max(b.robot_f(d, 'QT0'), b.robot_f(d, 'LOWER')), b.robot_f(d, 'UPPER'))
        v['q'][0, :nr] = q
        v['qd'][0] = 0
        v['qt'][0, :nr] = q
        hq_dyn = np.array([hq[j] for j in self.human_dyn])
        v['q'][0, nr:] = hq_dyn
        v['qt'][0, nr:] = hq_dyn
        v['tremor'][0] = tremors
        v['tremor_target'][0] = hq_dyn                                             # human.py:123 target_joint_angles
        # human.setup_joints(use_static_joints=True) after the settle (bed_bathing.py:134): every link static unless the impairment is tremor
        v['frozen'][0] = 0 if (impairment == 'tremor' or b.is_coop) else (((1 << nh) - 1) << nr)
        v['limit_scale'][0] = limit_scale
        v['base'][0, :3], v['base'][0, 3:] = rb_pos, rb_quat
        # tool in the gripper (tool.py:49-62): base frame = end-effector frame o TOOL; the record holds the COM frame
        if self.mount == 'mobile':
            tp, tq = self.mobile.kin.tool_pose(rb_pos, rb_quat, q)
        else:
            pe, Re, _, _ = self.arm.fk(rb_pos[None], X.quat_to_mat(rb_quat)[None], q_arm[None])
            tp, tq = X.compose(pe[0], X.mat_to_quat(Re[0]), b.task_f('TOOL_POS', 3), b.task_f('TOOL_QUAT', 4))
        ip, iq = X.invert(b.free_f(0, 'REFPOS', 3), b.free_f(0, 'REFQUAT', 4))
        cp, cq = X.compose(tp, tq, ip, iq)
        free = v['free'][0]
        free[:] = 0
        free[0, :3], free[0, 3:7] = cp, cq
        g = 0 if gender == 'male' else 1
        nt = b.task_i_n('NT', 4)[2 * g] + b.task_i_n('NT', 4)[2 * g + 1]
        v['plane_friction'][0] = plane_friction
        v['gender'][0] = g
        v['iteration'][0] = 0
        v['task_success'][0] = 0
        v['total_food'][0] = nt                                                    # total_target_count (bed_bathing.py:187)
        alive = np.zeros(L.BB['ALIVE_WORDS'], dtype=np.uint32)
        for t in range(nt):
            alive[t >> 5] |= np.uint32(1 << (t & 31))
        v['task'][0, L.BB['ALIVE']:L.BB['ALIVE'] + L.BB['ALIVE_WORDS']] = alive.view(np.int32)
        v['rng'][0, 0] = (env_seed * 2654435761 + 12345) & 0x7FFFFFFF
        v['rng'][0, 1] = (env_seed ^ 0x5bd1e995) & 0x7FFFFFFF
        if info is not None:
            info.update(gender=gender, impairment=impairment, limit_scale=limit_scale, strength=strength, tremors=tremors,
                        toc_goals=ngoal, toc_manipulability=manip, target_ee_pos=target_ee_pos, human_q=hq, human_base=(base_pos, base_quat))
        return state_row


def toc_search(blob):
    """the base pose search of reset_bed (BedBathingSawyerReset._toc) for another task's model: an object with ._toc(rng, start, goals)"""
    h = BedBathingSawyerReset.__new__(BedBathingSawyerReset)
    m = blob.meta
    h.blob, h.arm = blob, ArmChain(blob)
    h.toc_base = np.array([-0.85, -0.4, 0]) + np.array(m['toc_base'])          # robot.py:142 + toc_base_pos_offset
    h.ee_R = X.quat_to_mat(X.quat_from_rpy(m['ee_rpy']))                         # toc_ee_orient_rpy
    h.self_guard = m.get('robot') == 'sawyer'
    return h


def make_states(blob, n, seed=1001, impairment='random', settler=None, checker=None, **kw):
    """n independent post-reset states; env i uses RandomState(seed + i).  With a `settler` (a callable advancing bed_settle
    state records by the 100 simulation steps of bed_bathing.py:130-131, e.g. RagdollSettler) the human is settled as a rag
    doll, all environments in one batch; without one the rigid 'drop' stand-in is used.  checker(states) -> AGX_COLLIDE_* flags
    (DeviceCollisionChecker) turns on init_robot_pose's collision rejection (needs the settler path)."""
    rs = BedBathingSawyerReset(blob, settle='drop' if settler is None else 'ragdoll')
    st = blob.new_state(n)
    infos = [{} for _ in range(n)]
    rngs = [np.random.RandomState(seed + i) for i in range(n)]
    if settler is None:
        for i in range(n):
            rs.sample(rngs[i], st[i:i + 1], env_seed=seed + i, impairment=impairment, info=infos[i], **kw)
        return st, infos
    from ..blob import ModelBlob
    sblob = settler.blob if hasattr(settler, 'blob') else ModelBlob.load('bed_settle')
    pres = [rs.pre_settle(rngs[i], impairment=impairment, **kw) for i in range(n)]
    ss = sblob.new_state(n)
    for i, p in enumerate(pres):
        settle_record(sblob, ss[i:i + 1], p['gender'], p['limit_scale'], p['base_pos'], p['base_rpy'], p['hq'], p['plane_friction'])
    ss = settler(ss)
    for i, p in enumerate(pres):
        p['base_pos'], p['base_quat'], p['hq'] = settled_pose(sblob, ss[i:i + 1], p['hm'])
        rs.post_settle(rngs[i], st[i:i + 1], p, env_seed=seed + i, info=infos[i])
    if checker is not None:
        flags = reject_collisions(st, checker, lambda i, attempt: rs.post_settle(rngs[i], st[i:i + 1], pres[i], env_seed=seed + i, info=infos[i], attempt=attempt))
        for i in range(n):
            infos[i]['collision_flags'] = int(flags[i])
    return st, infos
