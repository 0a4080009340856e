"""Host-side (numpy, float64) kinematics of the compiled robot: forward kinematics, end-effector
Jacobian and a damped-least-squares IK used only at reset time, where the reference calls
``p.calculateInverseKinematics`` with random restarts (assistive_gym/envs/agents/robot.py:84-121,
agents/agent.py:252-274).  Bullet's IK internals are not reproduced: only the
acceptance test of ik_random_restarts (position / orientation thresholds) is kept."""
import numpy as np

from ..model import xform as X


class RobotKin:
    def __init__(self, blob):
        self.blob = blob
        n = blob.nrobot          # the robot's DoFs only (the human head chain is not part of the IK)
        self.n = n
        self.parent = [blob.robot_i(d, 'PARENT') for d in range(n)]
        self.tpos = [blob.robot_f(d, 'TPOS', 3) for d in range(n)]
        self.tquat = [blob.robot_f(d, 'TQUAT', 4) for d in range(n)]
        self.axis = [blob.robot_f(d, 'AXIS', 3) for d in range(n)]
        self.lower = np.array([blob.robot_f(d, 'LOWER') for d in range(n)])
        self.upper = np.array([blob.robot_f(d, 'UPPER') for d in range(n)])
        self.act = [blob.robot_i(d, 'ACT') for d in range(n)]
        self.prismatic = [blob.robot_i(d, 'JTYPE') == 1 for d in range(n)]
        self.arm = [d for d in range(n) if self.act[d] >= 0]
        self.arm.sort(key=lambda d: self.act[d])
        self.ee_link = blob.task_i('EE_LINK')
        self.ee_pos = blob.task_f('EE_POS', 3)
        self.ee_quat = blob.task_f('EE_QUAT', 4)
        self.tool_pos = blob.task_f('TOOL_POS', 3)
        self.tool_quat = blob.task_f('TOOL_QUAT', 4)

    def fk(self, base_pos, base_quat, q):
        pos, quat = [None] * self.n, [None] * self.n
        for d in range(self.n):
            pp, pq = (base_pos, base_quat) if self.parent[d] < 0 else (pos[self.parent[d]], quat[self.parent[d]])
            jp, jq = X.compose(pp, pq, self.tpos[d], self.tquat[d])
            if self.prismatic[d]:
                pos[d], quat[d] = jp + X.quat_rotate(jq, self.axis[d] * q[d]), jq
            else:
                pos[d], quat[d] = jp, X.quat_mul(jq, X.quat_from_axis_angle(self.axis[d], q[d]))
        return np.array(pos), np.array(quat)

    def ee_pose(self, base_pos, base_quat, q):
        pos, quat = self.fk(base_pos, base_quat, q)
        return X.compose(pos[self.ee_link], quat[self.ee_link], self.ee_pos, self.ee_quat)

    def tool_pose(self, base_pos, base_quat, q):
        """Tool.get_transform (agents/tool.py:49-58): end-effector frame o tool offset."""
        p, o = self.ee_pose(base_pos, base_quat, q)
        return X.compose(p, o, self.tool_pos, self.tool_quat)

    def ee_jacobian(self, base_pos, base_quat, q):
        pos, quat = self.fk(base_pos, base_quat, q)
        pe, _ = X.compose(pos[self.ee_link], quat[self.ee_link], self.ee_pos, self.ee_quat)
        J = np.zeros((6, self.n))
        d = self.ee_link
        while d >= 0:
            a = X.quat_rotate(quat[d], self.axis[d])
            if self.prismatic[d]:
                J[:3, d] = a
            else:
                J[:3, d] = np.cross(a, pe - pos[d])
                J[3:, d] = a
            d = self.parent[d]
        return J

    def ik(self, base_pos, base_quat, q0, target_pos, target_quat, iters=200, damping=0.05, lower=None, upper=None):
        """Damped least squares on the arm joints; returns the full joint vector."""
        q = np.array(q0, dtype=np.float64)
        lo = self.lower if lower is None else lower
        hi = self.upper if upper is None else upper
        arm = self.arm
        for _ in range(iters):
            p, o = self.ee_pose(base_pos, base_quat, q)
            ep = target_pos - p
            qe = X.quat_mul(target_quat, X.quat_conj(o))
            if qe[3] < 0:
                qe = -qe
            er = 2.0 * qe[:3]
            err = np.concatenate([ep, er])
            if np.linalg.norm(ep) < 1e-4 and np.linalg.norm(er) < 1e-4:
                break
            J = self.ee_jacobian(base_pos, base_quat, q)[:, arm]
            dq = J.T @ np.linalg.solve(J @ J.T + damping ** 2 * np.eye(6), err)
            step = np.max(np.abs(dq))
            if step > 0.5:
                dq *= 0.5 / step
            q[arm] = np.clip(q[arm] + dq, lo[arm], hi[arm])
        return q
