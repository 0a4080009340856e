"""Host-side reset for the feeding task (FeedingJaco-v1 and the other robots of model/compiler.py FEEDING_ROBOTS): produces the
pre-settle state record of one environment.  Wheelchair-mounted robots (Jaco, Panda) get Robot.ik_random_restarts from their fixed base;
free-standing ones (Sawyer, Baxter) the base pose search of Robot.position_robot_toc (host/reset_bed.py) with the goals
[(target_ee_pos, target_ee_orient)] + [(mouth, None)] (feeding.py:141).

Follows the order of FeedingEnv.reset (assistive_gym/envs/feeding.py:114-182) and what it calls:
build_assistive_env (envs/env.py:114-134: plane friction U(0.025,0.5)), Human.init
(agents/human.py:72-102: gender, impairment, limit scale, strength, tremors), create_human
(human_creation.py:58-66: skin colour draw), head angles U(-30,30)^3 deg (feeding.py:125),
Human.setup_joints (human.py:104-127), target end-effector position (feeding.py:139),
init_robot_pose -> Robot.ik_random_restarts (env.py:276-310, robot.py:84-121), gripper
(feeding.py:144), bowl offset (furniture.py:33), food grid (feeding.py:158-166).

Reset is outside the kernel scope; only its RESULT feeds the stepper.  Seed-level
parity with the reference through reset() is not attainable (the number of RNG draws depends on
Bullet's IK); the draw ORDER up to the IK call is kept.

The 25 settle steps of feeding.py:178-179 are run by the caller on the device (agx_settle).

Unlike the device-side generator (csrc/agx_reset.h, the default everywhere), this numpy sampler does NOT run the collision
rejection of robot.py:105-112 / env.py:299-308: it has no narrowphase of its own.  It is kept as an explicit alternative
(`reset='host'`) for tests and for seeds drawn in the reference's MT19937 order.
"""
import numpy as np

from ..model import xform as X
from ..model.human import HumanModel
from .kin import RobotKin

D = np.deg2rad


class MobilePlacement:
    """init_robot_pose for a robot on wheels (env.py:282-293): the base around toc_base_pos_offset (x, y +- 0.1), its yaw around
    toc_ee_orient_rpy's (+- 30 degrees; none of it in the dressing task), no IK; Stretch.randomize_init_joint_angles (stretch.py:58-62) draws the
    lift height.  The record's base pose is the anchor of the robot's six virtual joints, which start at zero."""

    def __init__(self, blob):
        self.blob, self.kin = blob, RobotKin(blob)
        m = blob.meta
        self.base, self.rpy, self.lift, self.lift_dof = np.array(m['mobile_base'], dtype=np.float64), np.array(m['mobile_rpy'], dtype=np.float64), m['lift'], m['lift_dof']
        self.yaw_range = D(30) if m.get('mobile_yaw', True) else 0.0

    def draw(self, prng):
        """-> base position, base quaternion, joint vector of the robot (virtual joints included)"""
        kin = self.kin
        pos = self.base.copy()
        pos[:2] += prng.uniform(-0.1, 0.1, size=2)
        rpy = self.rpy.copy()
        if self.yaw_range > 0:
            rpy[2] += prng.uniform(-self.yaw_range, self.yaw_range)
        q = np.clip(np.zeros(kin.n), kin.lower, kin.upper)                         # Agent.init -> enforce_joint_limits
        q[self.lift_dof] = self.lift + prng.uniform(-0.1, 0.1)
        return pos, X.quat_from_rpy(rpy), q


class FeedingJacoReset:
    def __init__(self, blob):
        self.blob = blob
        self.kin = RobotKin(blob)
        self.base_pos = np.array(blob.meta.get('robot_base_pos', [-0.35, -0.3, 0.36]), dtype=np.float64)
        self.base_quat = np.array(blob.meta.get('robot_base_quat', X.quat_from_rpy([0, 0, -np.pi / 2]).tolist()))
        self.human_bodies = blob.meta.get('human_bodies')
        self.human_dyn = blob.meta.get('human_dynamic_joints', [20, 21, 22, 23])
        self.toc_ee_orient = X.quat_from_rpy(blob.meta.get('ee_rpy', [np.pi / 2.0, 0, np.pi / 2.0]))     # toc_ee_orient_rpy (jaco.py:43)
        self.toc = None
        if blob.meta.get('mount', 'wheelchair') == 'toc':
            from .reset_bed import toc_search
            self.toc = toc_search(blob)
        self.mobile = MobilePlacement(blob) if blob.meta.get('mount') == 'mobile' else None
        self._hm_cache = {}

    def _human(self, gender, limit_scale):
        key = (gender, round(float(limit_scale), 9))
        if key not in self._hm_cache:
            if len(self._hm_cache) > 64:
                self._hm_cache.clear()
            self._hm_cache[key] = HumanModel(gender, limit_scale)
        return self._hm_cache[key]

    def sample(self, rng, state_row, env_seed=0, impairment='random', gender='random', max_restarts=1000, info=None, attempt=0):
        """Fill one state record (float32 view of length state_words) in place.  attempt > 0 (free-standing robots): a re-draw of the
        robot's placement only (init_robot_pose's rejection loop, env.py:281-308)."""
        b, kin = self.blob, self.kin
        v = b.view(state_row)
        plane_friction = rng.uniform(0.025, 0.5)                                   # env.py:120
        if gender not in ('male', 'female'):
            gender = rng.choice(['male', 'female'])                                # human.py:76-77
        if impairment == 'random':
            impairment = rng.choice(['none', 'limits', 'weakness', 'tremor'])      # human.py:80-81
        elif impairment == 'no_tremor':
            impairment = rng.choice(['none', 'limits', 'weakness'])
        limit_scale = 1.0 if impairment != 'limits' else rng.uniform(0.5, 1.0)     # human.py:85
        strength = 1.0 if impairment != 'weakness' else rng.uniform(0.25, 1.0)     # human.py:86
        tremors = np.zeros(4)
        if impairment == 'tremor':
            tremors = rng.uniform(D(-20), D(20), size=4)                           # human.py:89-90 (head joints)
        rng.uniform(0.4, 0.8)                                                      # skin colour, human_creation.py:63
        hm = self._human(gender, limit_scale)
        hq = hm.clamp(np.zeros(hm.n))                                              # human_creation.py:301-314
        jp = [(6, -90), (16, -90), (28, -90), (31, 80), (35, -90), (38, 80),       # feeding.py:124
              (21, rng.uniform(-30, 30)), (22, rng.uniform(-30, 30)), (23, rng.uniform(-30, 30))]   # feeding.py:125
        for j, a in jp:
            hq[j] = D(a)
        hq = hm.clamp(hq)                                                          # set_joint_angles(use_limits) + enforce_joint_limits
        hbase = np.array([0, 0.03, 0.89 if gender == 'male' else 0.86])            # human.py:102
        hpos, hquat = hm.fk(hbase, np.array([0, 0, 0, 1.0]), hq)
        for k, link in enumerate(self.human_bodies):
            if link < 0:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hbase, [0, 0, 0, 1]
            else:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hpos[link], hquat[link]
        mouth = b.task_f('MOUTH_M' if gender == 'male' else 'MOUTH_F', 3)
        target, _ = X.compose(hpos[23], hquat[23], mouth, np.array([0, 0, 0, 1.0]))   # feeding.py:184-196
        # robot start pose: IK with random restarts towards a point in front of the person
        target_ee_pos = np.array([-0.15, -0.65, 1.15]) + rng.uniform(-0.05, 0.05, size=3)      # feeding.py:139
        q = np.clip(np.zeros(kin.n), kin.lower, kin.upper)                         # Agent.init -> enforce_joint_limits
        ik_lo = np.where(kin.lower < -1e9, -2 * np.pi, kin.lower)                  # agent.py:223-231
        ik_hi = np.where(kin.upper > 1e9, 2 * np.pi, kin.upper)
        best, best_d, ok, restarts = None, np.inf, False, 0
        base_pos, base_quat = self.base_pos, self.base_quat
        if b.meta.get('mount') == 'mobile':
            # a robot on wheels (MobilePlacement).  attempt > 0: init_robot_pose's re-draw after a collision (env.py:299-308)
            from .reset_bed import placement_rng
            prng = placement_rng(np.random.RandomState(rng.randint(1 << 31)), env_seed, attempt)
            base_pos, base_quat, best = self.mobile.draw(prng)
            p, o = kin.ee_pose(base_pos, base_quat, best)
            best_d, ok, max_restarts = float(np.linalg.norm(target_ee_pos - p)), True, 0
        elif self.toc is not None:
            from .reset_bed import placement_rng
            prng = placement_rng(np.random.RandomState(rng.randint(1 << 31)), env_seed, attempt)
            res = None
            for _ in range(4):
                res = self.toc._toc(prng, target_ee_pos, [target])
                if res is not None:
                    break
            assert res is not None, 'no reachable base pose found'
            base_pos, base_quat, q_arm, ngoal, manip = res
            best = q.copy()
            for k_, d_ in enumerate(self.toc.arm.chain):
                best[d_] = q_arm[k_]
            p, o = kin.ee_pose(base_pos, base_quat, best)
            best_d, ok, max_restarts = float(np.linalg.norm(target_ee_pos - p)), True, 0
        for r in range(max_restarts):
            restarts = r + 1
            lo, hi = ik_lo, ik_hi
            if r >= 10:                                                            # robot.py:91 randomize_limits
                lo = rng.uniform(0, 1, size=kin.n) * ik_lo
                hi = rng.uniform(0, 1, size=kin.n) * ik_hi
            rest = rng.uniform(lo, hi)                                             # agent.py:263
            qs = kin.ik(self.base_pos, self.base_quat, rest, target_ee_pos, self.toc_ee_orient,
                        lower=np.minimum(lo, hi), upper=np.maximum(lo, hi))
            qs = np.clip(qs, kin.lower, kin.upper)                                 # set_joint_angles(use_limits=True)
            p, o = kin.ee_pose(self.base_pos, self.base_quat, qs)
            dpos = np.linalg.norm(target_ee_pos - p)
            dor = min(np.linalg.norm(self.toc_ee_orient - o), np.linalg.norm(self.toc_ee_orient + o))
            if dpos < best_d:
                best, best_d = qs, dpos
            if dpos < 0.01 and dor < 0.01:                                         # robot.py:97 success_threshold
                best, ok = qs, True
                break
        q = best.copy()
        for d in range(kin.n):                                                     # gripper, feeding.py:144 (set instantly)
            if kin.act[d] < 0:
                q[d] = min(max(b.robot_f(d, 'QT0'), kin.lower[d]), kin.upper[d])
        nr = b.nrobot
        v['q'][0, :nr] = q
        v['qd'][0] = 0
        v['qt'][0, :nr] = q       # motors hold the start pose until the first action (see module docstring)
        # human head joints: dynamic links, frozen (mass 0, human.py:104-110) unless the impairment is tremor
        hq_dyn = np.array([hq[j] for j in self.human_dyn])
        v['q'][0, nr:] = hq_dyn
        v['qt'][0, nr:] = hq_dyn
        v['tremor'][0] = tremors
        v['tremor_target'][0] = hq_dyn                                             # human.py:123 target_joint_angles
        # a controllable human keeps its controllable joints dynamic (human.py:108)
        v['frozen'][0] = 0 if (impairment == 'tremor' or b.is_coop) else (((1 << b.nhdof) - 1) << nr)
        v['limit_scale'][0] = limit_scale
        v['base'][0, :3], v['base'][0, 3:] = base_pos, base_quat
        # tool in the gripper (tool.py:49-62)
        tp, tq = kin.tool_pose(base_pos, base_quat, q)
        free = v['free'][0]
        free[:] = 0
        free[:, 6] = 1.0
        free[b.h['TOOL_BODY'], :3], free[b.h['TOOL_BODY'], 3:7] = tp, tq
        # bowl on the table (furniture.py:32-34); URDF base frame -> COM frame
        bowl_base = np.array([-0.15, -0.65, 0.75]) + np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), 0])
        refp, refq = b.free_f(1, 'REFPOS', 3), b.free_f(1, 'REFQUAT', 4)
        ip, iq = X.invert(refp, refq)
        cp, cq = X.compose(bowl_base, np.array([0, 0, 0, 1.0]), ip, iq)
        free[1, :3], free[1, 3:7] = cp, cq
        # food grid above the spoon (feeding.py:158-166)
        r_food, k = 0.005, 0
        for i in range(2):
            for j in range(2):
                for kk in range(2):
                    free[b.h['FOOD0'] + k, :3] = np.array([i * 2 * r_food - 0.005, j * 2 * r_food, kk * 2 * r_food + 0.01]) + tp
                    k += 1
        v['plane_friction'][0] = plane_friction
        v['gender'][0] = 0 if gender == 'male' else 1
        v['target'][0] = target
        v['food_alive'][0] = (1 << b.nfood) - 1
        v['food_active'][0] = (1 << b.nfood) - 1
        v['iteration'][0] = 0
        v['task_success'][0] = 0
        v['total_food'][0] = b.nfood
        v['rng'][0, 0] = (env_seed * 2654435761 + 12345) & 0x7FFFFFFF
        v['rng'][0, 1] = (env_seed ^ 0x5bd1e995) & 0x7FFFFFFF
        if info is not None:
            info.update(gender=gender, impairment=impairment, limit_scale=limit_scale, strength=strength, tremors=tremors,
                        ik_ok=ok, ik_restarts=restarts, ik_pos_err=best_d, target_ee_pos=target_ee_pos)
        return state_row


FeedingReset = FeedingJacoReset


def make_states(blob, n, seed=1001, impairment='random', checker=None, **kw):
    """n independent post-reset (pre-settle) states; env i uses RandomState(seed + i).  checker(states) -> AGX_COLLIDE_* flags
    (reset_bed.DeviceCollisionChecker): init_robot_pose's collision rejection for the free-standing robots (env.py:281-308)."""
    rs = FeedingJacoReset(blob)
    st = blob.new_state(n)
    infos = [{} for _ in range(n)]

    def draw(i, attempt=0):
        rs.sample(np.random.RandomState(seed + i), st[i:i + 1], env_seed=seed + i, impairment=impairment, info=infos[i], attempt=attempt, **kw)
    for i in range(n):
        draw(i)
    if checker is not None and (rs.toc is not None or blob.meta.get('mount') == 'mobile'):
        from .reset_bed import reject_collisions
        flags = reject_collisions(st, checker, draw)
        for i in range(n):
            infos[i]['collision_flags'] = int(flags[i])
    return st, infos
