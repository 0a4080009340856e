"""Host-side reset for DressingBaxter-v1 / DressingBaxterHuman-v1: post-reset state records and garments (the stepper's input).

Follows the order of DressingEnv.reset (assistive_gym/envs/dressing.py:112-198): build_assistive_env('wheelchair_left')
(envs/env.py:114-134: plane friction, Human.init draws, agents/human.py:72-102), motor gains 0.01 (:121), the seated human with
both elbows bent and the left shoulder abducted (:123-124; its left arm stays dynamic, held by a reactive PD of gain 0.01 and
force 1 x strength when the human is not controllable), the target end-effector pose (:130-133), init_robot_pose ->
Robot.position_robot_toc on the human's LEFT (right_side=False: env.py:298, robot.py:143) with oriented goals 10 cm above the
shoulder, elbow and wrist (:134), the gripper (:140), the garment loaded relative to the end effector (:146-153: every node is
shifted by cloth_offset = start_ee_pos - cloth_orig_pos) and hung from the massless attachment sphere at the end effector,
gravity -9.81 / 2 on the cloth while 50 simulation steps let it settle (:178-192), then -9.81.

The 50-step settle needs the stepper: `settler` advances (state records, garments) by n stepSimulation calls -- ClothSettler runs
it on the device (agx_settle); tests pass an oracle-backed one.  As in host/reset_bed.py: Bullet's IK is replaced by damped least
squares (Baxter's half_range IK limits, baxter.py:49, are not modelled) and init_robot_pose's collision loop is not run.
"""
import numpy as np

from ..model import compiler as L
from ..model import xform as X
from .reset_bed import ArmChain, BedBathingSawyerReset, placement_rng, reject_collisions

D = np.deg2rad


def cloth_x0(blob):
    """node positions of the garment for cloth_offset = 0 (float64 [NN, 3])"""
    oc = blob.h['OFF_CLOTH']
    nn = int(blob.i[oc + L.CL['NN']])
    o = oc + int(blob.i[oc + L.CL['OFF_X0']])
    return blob.f[o:o + 3 * nn].reshape(nn, 3).astype(np.float64)


def cloth_nodes(blob):
    return int(blob.i[blob.h['OFF_CLOTH'] + L.CL['NN']])


class DressingReset(BedBathingSawyerReset):
    def __init__(self, blob):
        assert blob.task_kind == L.TASK_DRESSING
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
        self.toc_base = np.array([-0.85, -0.4, 0]) + np.array(m.get('toc_base', [1.7, 0.7, 0.925]))      # robot.py:142 + toc_base_pos_offset (baxter.py:39)
        self.fixed_base = np.array([0, 0, 0.06]) + np.array(m.get('toc_base', [0, 0, 0]))                # wheelchair position + offset, rpy (0, 0, pi/2) (dressing.py:116-118)
        self.ee_R = X.quat_to_mat(X.quat_from_rpy(m.get('ee_rpy', [0, -np.pi / 2.0, 0])))                # toc_ee_orient_rpy['dressing'][0] (baxter.py:45)
        self.ee_R_shoulder = X.quat_to_mat(X.quat_from_rpy(m.get('ee_rpy_shoulder', [np.pi / 2.0, -np.pi / 2.0, 0])))   # [-1]
        self.self_guard = m.get('robot') == 'sawyer'
        self.x0 = cloth_x0(blob)
        self.cloth_orig_pos = np.array(blob.meta['cloth_orig_pos'])
        self._hm = {}

    def _human(self, gender, limit_scale):
        from ..model.human import HumanModel
        key = (gender, round(float(limit_scale), 9))
        if key not in self._hm:
            if len(self._hm) > 64:
                self._hm.clear()
            self._hm[key] = HumanModel(gender, limit_scale, cloth=True)
        return self._hm[key]

    def sample(self, rng, state_row, cloth_row, env_seed=0, impairment='random', gender='random', info=None, human_q_override=None, attempt=0):
        """Fill one state record and one garment (float32 [2, NN, 3]: positions, velocities) in place -- BEFORE the cloth has settled;
        the record's cloth gravity is the settle value -9.81 / 2 (dressing.py:178)."""
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
        hq = hm.clamp(np.zeros(hm.n))
        for j, a in [(6, -90), (13, -45), (16, -90), (28, -90), (31, 80), (35, -90), (38, 80)]:     # dressing.py:123
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
        shoulder, elbow, wrist = hpos[15], hpos[17], hpos[19]                      # left shoulder / elbow / wrist (dressing.py:126-128)
        target_ee_pos = np.array([0.45, -0.3, 1]) + rng.uniform(-0.05, 0.05, size=3)    # dressing.py:130
        off = np.array([0, 0, 0.1])
        toc = None
        rng = placement_rng(rng, env_seed, attempt)     # attempt > 0: a re-draw of the placement only (env.py:281); nothing else is drawn after it
        if self.mount == 'mobile':
            toc = self.mobile.draw(rng) + (0, 0.0)
        elif self.mount == 'wheelchair':     # Robot.ik_random_restarts from the fixed base on the human's left (env.py:295-297)
            toc = self._mounted_ik(rng, target_ee_pos, self.fixed_base, X.quat_from_rpy([0, 0, np.pi / 2.0]), human=(hm, hpos, hquat, hbase))
        else:
            for _ in range(4):
                toc = self._toc(rng, target_ee_pos, [shoulder + off, elbow + off, wrist + off], goal_Rs=[self.ee_R_shoulder, self.ee_R, self.ee_R], right_side=False)
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
        for d in range(nr):                                                        # gripper open position, set instantly (dressing.py:140)
            if b.robot_i(d, 'ACT') < 0:
                q[d] = min(max(b.robot_f(d, 'QT0'), b.robot_f(d, 'LOWER')), b.robot_f(d, 'UPPER'))
        v['q'][0, :nr] = q
        v['qd'][0] = 0
        v['qt'][0, :nr] = q
        hq_dyn = np.array([hq[j] for j in self.human_dyn])
        v['q'][0, nr:] = hq_dyn
        v['qt'][0, nr:] = hq_dyn
        v['tremor'][0] = tremors
        v['tremor_target'][0] = hq_dyn
        v['frozen'][0] = 0                                                         # the left arm stays dynamic (human.py:108,124-127)
        agent = b.is_coop or impairment == 'tremor'
        v['human_kp'][0] = 0.0 if agent else 0.01                                  # reactive_gain, dressing.py:124
        v['human_maxf'][0] = 0.0 if agent else 1.0 * strength
        v['limit_scale'][0] = limit_scale
        v['base'][0, :3], v['base'][0, 3:] = rb_pos, rb_quat
        if self.mount == 'mobile':
            start_ee_pos = self.mobile.kin.ee_pose(rb_pos, rb_quat, q)[0]
        else:
            pe, Re, _, _ = self.arm.fk(rb_pos[None], X.quat_to_mat(rb_quat)[None], q_arm[None])
            start_ee_pos = pe[0]                                                   # dressing.py:146
        cloth_offset = start_ee_pos - self.cloth_orig_pos                          # :148-149
        cloth_row[0] = (self.x0 + cloth_offset).astype(np.float32)
        cloth_row[1] = 0
        task = v['task'][0]
        task[:] = 0
        task[L.DR['CLOTH_GRAVITY']:L.DR['CLOTH_GRAVITY'] + 1] = np.array([-9.81 / 2], dtype=np.float32).view(np.int32)    # :178
        v['plane_friction'][0] = plane_friction
        v['gender'][0] = 0 if gender == 'male' else 1
        v['iteration'][0] = 0
        v['task_success'][0] = 0
        v['total_food'][0] = 1
        v['rng'][0, 0] = (env_seed * 2654435761 + 12345) & 0x7FFFFFFF
        v['rng'][0, 1] = (env_seed ^ 0x5bd1e995) & 0x7FFFFFFF
        if info is not None:
            info.update(gender=gender, impairment=impairment, limit_scale=limit_scale, strength=strength, tremors=tremors,
                        toc_goals=ngoal, toc_manipulability=manip, target_ee_pos=target_ee_pos, human_q=hq, start_ee_pos=start_ee_pos)
        return state_row


def finish_settle(blob, states):
    """after the settle: full gravity on the cloth (dressing.py:195), velocities of the articulated bodies as they are"""
    v = blob.view(states)
    v['task'][:, L.DR['CLOTH_GRAVITY']] = np.array([-9.81], dtype=np.float32).view(np.int32)[0]
    v['iteration'][:] = 0
    return states


class ClothSettler:
    """Runs the 50-step cloth settle of DressingEnv.reset on the device (agx_settle on the dressing model with the garments attached)."""

    def __init__(self, blob, n_envs, device=0):
        from ..libagx import Stepper
        self.blob = blob
        self.ctx = Stepper(blob, n_envs, device)
        self.n = n_envs

    def __call__(self, states, cloth, n_sim_steps):
        n = len(states)
        assert n <= self.n
        sb, cb = self.blob.new_state(self.n), np.zeros((self.n,) + cloth.shape[1:], dtype=np.float32)
        sb[:n], cb[:n] = states, cloth
        sb[n:], cb[n:] = states[:1], cloth[:1]
        self.ctx.set_state(sb)
        self.ctx.set_cloth(cb)
        self.ctx.settle(n_sim_steps)
        self.ctx.L.agx_synchronize(self.ctx.h, None)
        return self.ctx.get_state()[:n], self.ctx.get_cloth()[:n]


def make_states(blob, n, seed=1001, impairment='random', settler=None, settle_steps=50, checker=None, **kw):
    """n independent post-reset (state record, garment) pairs; env i uses RandomState(seed + i).  settler(states, cloth, n_sim_steps)
    -> (states, cloth) runs the cloth settle (ClothSettler: on the device); without one the garment is left as loaded."""
    rs = DressingReset(blob)
    st = blob.new_state(n)
    cloth = np.zeros((n, 2, cloth_nodes(blob), 3), dtype=np.float32)
    infos = [{} for _ in range(n)]

    def draw(i, attempt=0):
        rs.sample(np.random.RandomState(seed + i), st[i:i + 1], cloth[i], env_seed=seed + i, impairment=impairment, info=infos[i], attempt=attempt, **kw)
    for i in range(n):
        draw(i)
    if checker is not None:          # init_robot_pose's collision rejection (env.py:281-308), before the garment is loaded
        flags = reject_collisions(st, checker, draw)
        for i in range(n):
            infos[i]['collision_flags'] = int(flags[i])
    if settler is not None:
        st, cloth = settler(st, cloth, settle_steps)
    return finish_settle(blob, st), cloth, infos


DressingBaxterReset = DressingReset
