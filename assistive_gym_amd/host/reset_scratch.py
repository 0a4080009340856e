"""Host-side reset for ScratchItch<Robot>-v1 / ScratchItch<Robot>Human-v1 (PR2, Sawyer: base pose search; Jaco, Panda: mounted on the
wheelchair): produces post-reset state records (the stepper's input).

Follows the order of ScratchItchEnv.reset (assistive_gym/envs/scratch_itch.py:93-132): build_assistive_env('wheelchair')
(envs/env.py:114-134: plane friction, Human.init draws, agents/human.py:72-102), the seated human with its joint presets
(scratch_itch.py:104-105; Human.setup_joints, human.py:104-127: the right arm stays dynamic -- held by a reactive PD of gain 0.01
and force 1 x strength when the human is not controllable), the target end-effector pose (:115-116), init_robot_pose ->
Robot.position_robot_toc (env.py:276-310, robot.py:123-215) for the base and arm of a free-standing robot, or Robot.ik_random_restarts
(env.py:295-297, robot.py:84-121) from the fixed base of a wheelchair-mounted one (scratch_itch.py:97-99), the gripper (:120), generate_target
(:134-146: limb draw + Util.point_on_capsule, util.py:58-78).

As in host/reset_bed.py (whose TOC search and batched IK this reuses): Bullet's IK is replaced by damped least squares and the
collision rejection loop of init_robot_pose is not run.
"""
import numpy as np

from ..model import compiler as L
from ..model import xform as X
from ..model.human import HumanModel
from .reset_bed import ArmChain, BedBathingSawyerReset, mat_to_quat_batch, placement_rng, reject_collisions

D = np.deg2rad


def point_on_limb(rng, radius, length):
    """Util.point_on_capsule (util.py:58-78) for p1 = 0, p2 = (0, 0, -length), theta_range = (0, 2 pi), as generate_target calls it
    (scratch_itch.py:140): two draws, a length along the axis in [radius, length] and an angle."""
    rl = rng.uniform(radius, length)
    th = rng.uniform(0, 2 * np.pi)
    axis, ortho, normal = np.array([0, 0, -1.0]), np.array([0, -1.0, 0]), np.array([-1.0, 0, 0])     # Util.orthogonal_vector of (0, 0, -1)
    return rl * axis + radius * np.cos(th) * ortho + radius * np.sin(th) * normal


class ScratchItchReset(BedBathingSawyerReset):
    def __init__(self, blob):
        assert blob.task_kind == L.TASK_SCRATCH_ITCH
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
        self.toc_base = np.array([-0.85, -0.4, 0]) + np.array(m.get('toc_base', [0.1, 0, 0]))       # robot.py:142 + toc_base_pos_offset (pr2.py:35)
        self.fixed_base = np.array([0, 0, 0.06]) + np.array(m.get('toc_base', [0, 0, 0]))          # wheelchair position + offset (scratch_itch.py:97-99)
        self.ee_R = X.quat_to_mat(X.quat_from_rpy(m.get('ee_rpy', [0, 0, 0])))                      # toc_ee_orient_rpy (pr2.py:41)
        self.self_guard = m.get('robot') == 'sawyer'                                                 # see reset_bed._arm_in_pedestal
        self._hm = {}

    def sample(self, rng, state_row, env_seed=0, impairment='random', gender='random', info=None, human_q_override=None, attempt=0):
        """attempt > 0: a re-draw of the robot's placement only (init_robot_pose's rejection loop), everything else as on attempt 0"""
        b = self.blob
        v = b.view(state_row)
        nr, nh = b.nrobot, b.nhdof
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
            tremors = rng.uniform(D(-10), D(10), size=nh)                          # human.py:91-92
        rng.uniform(0.4, 0.8)                                                      # skin colour, human_creation.py:63
        hm = self._human(gender, limit_scale)
        hq = hm.clamp(np.zeros(hm.n))                                              # human_creation.py:301-314
        for j, a in [(3, 30), (6, -90), (16, -90), (28, -90), (31, 80), (35, -90), (38, 80)]:     # scratch_itch.py:104
            hq[j] = D(a)
        for j, a in (human_q_override or {}).items():                              # tests only
            hq[j] = a
        hq = hm.clamp(hq)
        hbase = np.array([0, 0.03, 0.89 if gender == 'male' else 0.86])            # human.py:102
        hpos, hquat = hm.fk(hbase, np.array([0, 0, 0, 1.0]), hq)
        for k, link in enumerate(self.human_bodies):
            if link < 0:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hbase, [0, 0, 0, 1]
            else:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hpos[link], hquat[link]
        shoulder, elbow, wrist = hpos[5], hpos[7], hpos[9]                         # scratch_itch.py:107-109
        target_ee_pos = np.array([-0.6, 0, 0.8]) + rng.uniform(-0.05, 0.05, size=3)    # scratch_itch.py:115
        toc = None
        prng = placement_rng(np.random.RandomState(rng.randint(1 << 31)), env_seed, attempt)     # one draw of the main stream, whatever the attempt
        if self.mount == 'mobile':
            toc = self.mobile.draw(prng) + (0, 0.0)
        elif self.mount == 'wheelchair':
            toc = self._mounted_ik(prng, target_ee_pos, self.fixed_base, X.quat_from_rpy([0, 0, -np.pi / 2.0]), human=(hm, hpos, hquat, hbase))   # scratch_itch.py:99
        else:
            for _ in range(4):
                toc = self._toc(prng, target_ee_pos, [shoulder, elbow, wrist])
                if toc is not None:
                    break
        assert toc is not None, 'no reachable base pose found'
        rb_pos, rb_quat, q_arm, ngoal, manip = toc
        if self.mount == 'mobile':
            q = q_arm.copy()
        else:
            q = np.zeros(nr)
            for k, d in enumerate(self.arm.chain):
                q[d] = q_arm[k]
        for d in range(nr):                                                        # gripper open position, set instantly (scratch_itch.py:120)
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
        # the controllable joints stay dynamic: controllable, or held by the reactive PD (reactive_force = 1, human.py:108,124-127)
        v['frozen'][0] = 0
        agent = b.is_coop or impairment == 'tremor'                                # then take_step re-sets the motors every step (env.py:130-131,222)
        v['human_kp'][0] = 0.0 if agent else 0.01                                  # reactive_gain, scratch_itch.py:105
        v['human_maxf'][0] = 0.0 if agent else 1.0 * strength
        v['limit_scale'][0] = limit_scale
        v['base'][0, :3], v['base'][0, 3:] = rb_pos, rb_quat
        if self.mount == 'mobile':
            tp, tq = self.mobile.kin.tool_pose(rb_pos, rb_quat, q)
        else:
            pe, Re, _, _ = self.arm.fk(rb_pos[None], X.quat_to_mat(rb_quat)[None], q_arm[None])
            tp, tq = X.compose(pe[0], X.mat_to_quat(Re[0]), b.task_f('TOOL_POS', 3), b.task_f('TOOL_QUAT', 4))      # tool.py:49-62
        ip, iq = X.invert(b.free_f(0, 'REFPOS', 3), b.free_f(0, 'REFQUAT', 4))
        cp, cq = X.compose(tp, tq, ip, iq)
        free = v['free'][0]
        free[:] = 0
        free[0, :3], free[0, 3:7] = cp, cq
        # generate_target (scratch_itch.py:134-146): limb, then a point on its capsule (util.py:58-78)
        limb = int(rng.randint(2))
        radius, length = hm.dims['upperarm' if limb == 0 else 'forearm']
        target_on_arm = point_on_limb(rng, radius, length)
        task = v['task'][0]
        task[:] = 0
        task[L.SI['TARGET']:L.SI['TARGET'] + 3] = target_on_arm.astype(np.float32).view(np.int32)
        task[L.SI['LIMB']] = limb
        v['plane_friction'][0] = plane_friction
        v['gender'][0] = 0 if gender == 'male' else 1
        v['iteration'][0] = 0
        v['task_success'][0] = 0
        v['total_food'][0] = 1                                                      # task_success >= 1 x task_success_threshold (scratch_itch.py:37)
        v['rng'][0, 0] = (env_seed * 2654435761 + 12345) & 0x7FFFFFFF
        v['rng'][0, 1] = (env_seed ^ 0x5bd1e995) & 0x7FFFFFFF
        if info is not None:
            info.update(gender=gender, impairment=impairment, limit_scale=limit_scale, strength=strength, tremors=tremors, limb=limb,
                        toc_goals=ngoal, toc_manipulability=manip, target_ee_pos=target_ee_pos, human_q=hq, target_on_arm=target_on_arm)
        return state_row


def make_states(blob, n, seed=1001, impairment='random', checker=None, **kw):
    """n independent post-reset states; env i uses RandomState(seed + i).  checker(states) -> AGX_COLLIDE_* flags
    (reset_bed.DeviceCollisionChecker) turns on init_robot_pose's collision rejection (env.py:281-308)."""
    rs = ScratchItchReset(blob)
    st = blob.new_state(n)
    infos = [{} for _ in range(n)]

    def draw(i, attempt=0):
        rs.sample(np.random.RandomState(seed + i), st[i:i + 1], env_seed=seed + i, impairment=impairment, info=infos[i], attempt=attempt, **kw)
    for i in range(n):
        draw(i)
    if checker is not None:
        flags = reject_collisions(st, checker, draw)
        for i in range(n):
            infos[i]['collision_flags'] = int(flags[i])
    return st, infos


ScratchItchPR2Reset = ScratchItchReset
