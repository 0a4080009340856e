"""Host-side reset for ArmManipulationSawyer-v1 / ArmManipulationSawyerHuman-v1: post-reset state records (the stepper's input).

Follows the order of ArmManipulationEnv.reset (assistive_gym/envs/arm_manipulation.py:110-182): build_assistive_env('bed',
fixed_human_base=False, human_impairment='no_tremor') (envs/env.py:114-134), motor forces 20 / 2 (:114-115), the posed human dropped
from [-0.25, 0.2, 0.95] as a rag doll for 100 simulation steps at gravity -1 onto the bed at friction 5 (:117-134), bed friction 0.3
(:136), the right arm posed at shoulder (60, -60) degrees / elbow 0 and left dynamic while every other link becomes static
(:139-142; a reactive hold of 0.01 N m x strength, gain 0.05), 100 more simulation steps in which the arm falls beside the body
(:145-146), the target end-effector poses (:158-159: both are drawn), init_robot_pose -> Robot.position_robot_toc with the goals wrist,
waist, elbow, stomach (:162), the scooper in the gripper (:154,173), gravity -9.81 with none on the tool (:176-177).

Both settles need the stepper: `settler` is the rag-doll settle of host/reset_bed.py (RagdollSettler), `arm_settler` advances
arm-manipulation state records by n stepSimulation calls at gravity -1 (ArmFallSettler: on the device, the robot and its tool parked
out of reach); tests pass oracle-backed ones.  As in host/reset_bed.py: Bullet's IK is replaced by damped least squares and
init_robot_pose's collision loop is not run.
"""
import numpy as np

from ..model import compiler as L
from ..model import xform as X
from .reset_bed import ArmChain, BedBathingSawyerReset, settle_record, settled_pose, placement_rng, reject_collisions

D = np.deg2rad
PARKED = np.array([20.0, 20.0, 0.975])          # where the robot stands while the arm falls (it is not yet placed then, :162)


class ArmManipulationSawyerReset(BedBathingSawyerReset):
    def __init__(self, blob):
        assert blob.task_kind == L.TASK_ARM_MANIPULATION
        self.blob = blob
        self.arm = ArmChain(blob)
        self.dual = bool(blob.meta.get('dual'))          # a two-armed robot: tool_right in the right hand (self.arm), tool_left in the left (self.arm2)
        self.arm2 = ArmChain(blob, second=True) if self.dual else None
        self.human_bodies = blob.meta['human_bodies']
        self.human_dyn = blob.meta['human_dynamic_joints']
        m = blob.meta
        self.toc_base = np.array([-0.85, -0.4, 0]) + np.array(m.get('toc_base', [-0.3, 0.6, 0.975]))     # robot.py:142 + toc_base_pos_offset (sawyer.py:40)
        self.ee_R = X.quat_to_mat(X.quat_from_rpy(m.get('ee_rpy', [0, -np.pi / 2.0, np.pi])))            # toc_ee_orient_rpy (sawyer.py:46)
        self.self_guard = m.get('robot', 'sawyer') == 'sawyer'                                            # see _arm_in_pedestal
        self._hm = {}

    def pre_settle(self, rng, impairment='no_tremor', gender='random', human_q_override=None):
        pre = BedBathingSawyerReset.pre_settle(self, rng, impairment, gender, human_q_override)
        pre['base_pos'] = np.array([-0.25, 0.2, 0.95])                                       # arm_manipulation.py:123
        return pre

    def _fill_human(self, v, pre):
        hm, hq, base_pos, base_quat = pre['hm'], pre['hq'], pre['base_pos'], pre['base_quat']
        hpos, hquat = hm.fk(base_pos, base_quat, hq)
        for k, link in enumerate(self.human_bodies):
            if link < 0:
                v['human'][0, k, :3], v['human'][0, k, 3:] = base_pos, base_quat
            else:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hpos[link], hquat[link]
        return hpos

    def _place(self, v, rb_pos, rb_quat, q_arm, q_arm2=None):
        """robot joints, base and the tool(s) in the gripper(s) (tool.py:49-62)"""
        b, nr = self.blob, self.blob.nrobot
        q = np.zeros(nr)
        for k, d in enumerate(self.arm.chain):
            q[d] = q_arm[k]
        if self.dual:
            for k, d in enumerate(self.arm2.chain):
                q[d] = q_arm2[k]
        for d in range(nr):                                                        # gripper open position, set instantly (:170)
            if b.robot_i(d, 'ACT') < 0:
                q[d] = min(max(b.robot_f(d, 'QT0'), b.robot_f(d, 'LOWER')), b.robot_f(d, 'UPPER'))
        v['q'][0, :nr] = q
        v['qd'][0, :nr] = 0
        v['qt'][0, :nr] = q
        v['base'][0, :3], v['base'][0, 3:] = rb_pos, rb_quat
        pe, Re, _, _ = self.arm.fk(np.asarray(rb_pos)[None], X.quat_to_mat(rb_quat)[None], q_arm[None])
        tp, tq = X.compose(pe[0], X.mat_to_quat(Re[0]), b.task_f('TOOL_POS', 3), b.task_f('TOOL_QUAT', 4))
        ip, iq = X.invert(b.free_f(0, 'REFPOS', 3), b.free_f(0, 'REFQUAT', 4))
        cp, cq = X.compose(tp, tq, ip, iq)
        free = v['free'][0]
        free[:] = 0
        free[0, :3], free[0, 3:7] = cp, cq
        if self.dual:
            pe, Re, _, _ = self.arm2.fk(np.asarray(rb_pos)[None], X.quat_to_mat(rb_quat)[None], q_arm2[None])
            tp, tq = X.compose(pe[0], X.mat_to_quat(Re[0]), b.task_f('TOOL2_POS', 3), b.task_f('TOOL2_QUAT', 4))
            ip, iq = X.invert(b.free_f(1, 'REFPOS', 3), b.free_f(1, 'REFQUAT', 4))
            free[1, :3], free[1, 3:7] = X.compose(tp, tq, ip, iq)

    def arm_fall_record(self, state_row, pre, env_seed=0):
        """the record the second settle starts from (:136-142): the human resting, its right arm posed, the robot parked"""
        b = self.blob
        v = b.view(state_row)
        state_row[:] = 0
        nr = b.nrobot
        hm, hq = pre['hm'], pre['hq']
        hq = hq.copy()
        hq[3], hq[4], hq[6] = D(60), D(-60), 0.0                                   # j_right_shoulder_x / _y, j_right_elbow (:139)
        hq = hm.clamp(hq)                                                          # enforce_joint_limits (human.py:121)
        pre['hq'] = hq
        self._fill_human(v, pre)
        lo, hi = self.arm.lower, self.arm.upper                                     # parked at the middle of its joint ranges (the Jaco's zero pose violates its limits)
        mid2 = None
        if self.dual:
            lo2, hi2 = self.arm2.lower, self.arm2.upper
            mid2 = np.where((lo2 > -1e9) & (hi2 < 1e9), 0.5 * (lo2 + hi2), 0.0)
        self._place(v, PARKED, np.array([0, 0, 0, 1.0]), np.where((lo > -1e9) & (hi < 1e9), 0.5 * (lo + hi), 0.0), mid2)
        hq_dyn = np.array([hq[j] for j in self.human_dyn])
        v['q'][0, nr:] = hq_dyn
        v['qt'][0, nr:] = hq_dyn
        v['qd'][0, nr:] = 0
        v['tremor'][0] = 0
        v['tremor_target'][0] = hq_dyn                                             # human.py:123 target_joint_angles
        v['frozen'][0] = 0                                                         # the right arm is dynamic in every episode
        # the reactive hold of setup_joints (:140, human.py:124-127); a controllable human's motors are re-targeted by every take_step
        v['human_kp'][0] = 0.0 if b.is_coop else 0.05
        v['human_maxf'][0] = 0.0 if b.is_coop else 0.01 * pre['strength']
        v['limit_scale'][0] = pre['limit_scale']
        v['plane_friction'][0] = pre['plane_friction']
        v['gender'][0] = 0 if pre['gender'] == 'male' else 1
        v['total_food'][0] = 1
        v['rng'][0, 0] = (env_seed * 2654435761 + 12345) & 0x7FFFFFFF
        v['rng'][0, 1] = (env_seed ^ 0x5bd1e995) & 0x7FFFFFFF
        return state_row

    def post_fall(self, rng, state_row, pre, env_seed=0, info=None, attempt=0):
        """everything after the arm has fallen (:148-179); the record keeps the arm's joint angles AND velocities"""
        b = self.blob
        v = b.view(state_row)
        nr = b.nrobot
        hm, hq = pre['hm'], pre['hq'].copy()
        for k, j in enumerate(self.human_dyn):
            hq[j] = v['q'][0, nr + k]
        hpos, _ = hm.fk(pre['base_pos'], pre['base_quat'], hq)
        elbow, wrist, stomach, waist = hpos[7], hpos[9], hpos[24], hpos[27]        # :148-151
        if attempt == 0:
            pre['target_ee_pos'] = np.array([-1, -0.3 if self.dual else 0.4, 0.8]) + rng.uniform(-0.05, 0.05, size=3)    # :158
            pre['target_ee_left_pos'] = np.array([-1, 0.7, 0.8]) + rng.uniform(-0.05, 0.05, size=3)                    # :159 (drawn for a single arm as well)
        target_ee_pos = pre['target_ee_pos']
        prng = placement_rng(rng, env_seed, attempt)
        toc, q_arm2 = None, None
        for _ in range(4):
            if self.dual:     # :165: the right arm's goals are wrist and waist, the left arm's elbow and stomach
                toc = self._toc_dual(prng, [self.arm, self.arm2], [target_ee_pos, pre['target_ee_left_pos']], [[wrist, waist], [elbow, stomach]])
            else:
                toc = self._toc(prng, target_ee_pos, [wrist, waist, elbow, stomach])   # :162
            if toc is not None:
                break
        assert toc is not None, 'no reachable base pose found'
        rb_pos, rb_quat, q_arm, ngoal, manip = toc
        if self.dual:
            q_arm, q_arm2 = q_arm
        self._place(v, rb_pos, rb_quat, q_arm, q_arm2)
        v['iteration'][0] = 0
        v['task_success'][0] = 0
        v['task'][0] = 0                                                           # AM_BEST: task_success = 0 (init_env_variables)
        if info is not None:
            info.update(gender=pre['gender'], impairment=pre['impairment'], limit_scale=pre['limit_scale'], strength=pre['strength'],
                        toc_goals=ngoal, toc_manipulability=manip, target_ee_pos=target_ee_pos, human_q=hq, human_base=(pre['base_pos'], pre['base_quat']))
        return state_row


class ArmFallSettler:
    """Runs the second settle of ArmManipulationEnv.reset on the device: agx_settle on the arm-manipulation model with the human's gravity
    set to the reset's -1 (arm_manipulation.py:125,145-146).  Fails loudly without a GPU."""

    def __init__(self, blob, n_envs, device=0):
        from ..libagx import Stepper
        self.blob = blob.set_param('HUMAN_GRAVITY_Z', -1.0)
        self.ctx = Stepper(self.blob, n_envs, device)
        self.n = n_envs

    def __call__(self, states, n_sim_steps):
        n = len(states)
        assert n <= self.n
        buf = self.blob.new_state(self.n)
        buf[:n] = states
        buf[n:] = states[:1]
        self.ctx.set_state(buf)
        self.ctx.settle(n_sim_steps)
        self.ctx.L.agx_synchronize(self.ctx.h, None)
        return self.ctx.get_state()[:n]


def make_states(blob, n, seed=1001, impairment='no_tremor', settler=None, arm_settler=None, fall_steps=100, checker=None, **kw):
    """n independent post-reset states; env i uses RandomState(seed + i).  settler: bed_settle records -> records after the 100-step
    rag-doll settle (host/reset_bed.RagdollSettler; None = the rigid 'drop' stand-in); arm_settler(states, n_sim_steps): the arm's fall
    (ArmFallSettler; None = the arm stays as posed); checker(states) -> AGX_COLLIDE_* flags (reset_bed.DeviceCollisionChecker) turns on
    init_robot_pose's collision rejection (env.py:281-308)."""
    rs = ArmManipulationSawyerReset(blob)
    st = blob.new_state(n)
    infos = [{} for _ in range(n)]
    rngs = [np.random.RandomState(seed + i) for i in range(n)]
    pres = [rs.pre_settle(rngs[i], impairment=impairment, **kw) for i in range(n)]
    if settler is None:
        r = blob.meta['ranges']
        rs._bed = [blob.collider(c)['verts'] for c in range(*r['bed'])]
        rs._bed_box = np.array([[v.min(0), v.max(0)] for v in rs._bed])
        for p in pres:
            p['base_pos'] = rs._drop(p['hm'], p['base_pos'], p['base_quat'], p['hq'])
    else:
        from ..blob import ModelBlob
        sblob = settler.blob if hasattr(settler, 'blob') else ModelBlob.load('bed_settle')
        ss = sblob.new_state(n)
        for i, p in enumerate(pres):
            settle_record(sblob, ss[i:i + 1], p['gender'], p['limit_scale'], p['base_pos'], p['base_rpy'], p['hq'], p['plane_friction'])
        ss = settler(ss)
        for i, p in enumerate(pres):
            p['base_pos'], p['base_quat'], p['hq'] = settled_pose(sblob, ss[i:i + 1], p['hm'])
    for i, p in enumerate(pres):
        rs.arm_fall_record(st[i:i + 1], p, env_seed=seed + i)
    if arm_settler is not None:
        st = np.ascontiguousarray(arm_settler(st, fall_steps))
    for i, p in enumerate(pres):
        rs.post_fall(rngs[i], st[i:i + 1], p, env_seed=seed + i, info=infos[i])
    if checker is not None:
        flags = reject_collisions(st, checker, lambda i, attempt: rs.post_fall(rngs[i], st[i:i + 1], pres[i], env_seed=seed + i, info=infos[i], attempt=attempt))
        for i in range(n):
            infos[i]['collision_flags'] = int(flags[i])
    return st, infos
