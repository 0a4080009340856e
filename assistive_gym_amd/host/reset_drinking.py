"""Host-side reset for DrinkingJaco-v1 (MODEL + CPU ORACLE ONLY so far: no kernel variant serves the drinking task): the pre-settle
state record and the water buffer of one environment.

Follows the order of DrinkingEnv.reset (assistive_gym/envs/drinking.py:123-181): build_assistive_env('wheelchair') (env.py:114-134), the
mounted robot on the wheelchair (:126-128), motor gains 0.005 (:130), the seated human with the head draws (:132-134), generate_target
(:183-196), the cup in the gripper (:137, tool.py:49-62), target_ee_pos around [-0.2, -0.5, 1.1] and Robot.ik_random_restarts (:141-143,
env.py:295-297), the gripper (:146), the 4 x 4 x 4 grid of water spheres above the cup's base frame (:160-167).  The 50 settle steps of
:176-177 are the caller's (oracle: settle_cloth).  As in host/reset.py: Bullet's IK is replaced by damped least squares; no collision
rejection here.
"""
import numpy as np

from ..model import compiler as L
from ..model import xform as X
from .reset import FeedingJacoReset

D = np.deg2rad


def water_x0(blob):
    """rest offsets of the particles relative to the cup's base frame position (the section's X0)"""
    oc = blob.h['OFF_CLOTH']
    nn = int(blob.i[oc + L.CL['NN']])
    o = oc + int(blob.i[oc + L.CL['OFF_X0']])
    return blob.f[o:o + 3 * nn].reshape(nn, 3).astype(np.float64)


class DrinkingReset(FeedingJacoReset):
    def __init__(self, blob):
        assert blob.task_kind == L.TASK_DRINKING
        FeedingJacoReset.__init__(self, blob)
        self.x0 = water_x0(blob)

    def sample(self, rng, state_row, water_row, env_seed=0, impairment='random', gender='random', max_restarts=1000, info=None):
        b, kin = self.blob, self.kin
        v = b.view(state_row)
        plane_friction = rng.uniform(0.025, 0.5)                                   # env.py:120
        if gender not in ('male', 'female'):
            gender = rng.choice(['male', 'female'])                                # human.py:76-77
        if impairment == 'random':
            impairment = rng.choice(['none', 'limits', 'weakness', 'tremor'])      # human.py:80-81
        limit_scale = 1.0 if impairment != 'limits' else rng.uniform(0.5, 1.0)     # human.py:85
        strength = 1.0 if impairment != 'weakness' else rng.uniform(0.25, 1.0)     # human.py:86
        tremors = np.zeros(4)
        if impairment == 'tremor':
            tremors = rng.uniform(D(-20), D(20), size=4)                           # human.py:89-90 (head joints)
        rng.uniform(0.4, 0.8)                                                      # skin colour, human_creation.py:63
        hm = self._human(gender, limit_scale)
        hq = hm.clamp(np.zeros(hm.n))
        for j, a in [(6, -90), (16, -90), (28, -90), (31, 80), (35, -90), (38, 80),                           # drinking.py:132
                     (21, rng.uniform(-30, 30)), (22, rng.uniform(-30, 30)), (23, rng.uniform(-30, 30))]:     # :133
            hq[j] = D(a)
        hq = hm.clamp(hq)
        hbase = np.array([0, 0.03, 0.89 if gender == 'male' else 0.86])            # human.py:102
        hpos, hquat = hm.fk(hbase, np.array([0, 0, 0, 1.0]), hq)
        for k, link in enumerate(self.human_bodies):
            if link < 0:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hbase, [0, 0, 0, 1]
            else:
                v['human'][0, k, :3], v['human'][0, k, 3:] = hpos[link], hquat[link]
        mouth = b.task_f('MOUTH_M' if gender == 'male' else 'MOUTH_F', 3)
        target, _ = X.compose(hpos[23], hquat[23], mouth, np.array([0, 0, 0, 1.0]))   # drinking.py:190-196
        target_ee_pos = np.array([-0.2, -0.5, 1.1]) + rng.uniform(-0.05, 0.05, size=3)          # drinking.py:141
        q = np.clip(np.zeros(kin.n), kin.lower, kin.upper)
        ik_lo = np.where(kin.lower < -1e9, -2 * np.pi, kin.lower)                  # agent.py:223-231
        ik_hi = np.where(kin.upper > 1e9, 2 * np.pi, kin.upper)
        best, best_d, ok, restarts = q.copy(), np.inf, False, 0
        for r in range(max_restarts):
            restarts = r + 1
            lo, hi = ik_lo, ik_hi
            if r >= 10:                                                            # robot.py:91 randomize_limits
                lo = rng.uniform(0, 1, size=kin.n) * ik_lo
                hi = rng.uniform(0, 1, size=kin.n) * ik_hi
            rest = rng.uniform(lo, hi)                                             # agent.py:263
            qs = kin.ik(self.base_pos, self.base_quat, rest, target_ee_pos, self.toc_ee_orient, lower=np.minimum(lo, hi), upper=np.maximum(lo, hi))
            qs = np.clip(qs, kin.lower, kin.upper)
            p, o = kin.ee_pose(self.base_pos, self.base_quat, qs)
            dpos = np.linalg.norm(target_ee_pos - p)
            dor = min(np.linalg.norm(self.toc_ee_orient - o), np.linalg.norm(self.toc_ee_orient + o))
            if dpos < best_d:
                best, best_d = qs, dpos
            if dpos < 0.01 and dor < 0.01:                                         # robot.py:97
                best, ok = qs, True
                break
        q = best.copy()
        for d in range(kin.n):                                                     # gripper, drinking.py:146 (set instantly)
            if kin.act[d] < 0:
                q[d] = min(max(b.robot_f(d, 'QT0'), kin.lower[d]), kin.upper[d])
        nr = b.nrobot
        v['q'][0, :nr], v['qd'][0], v['qt'][0, :nr] = q, 0, q
        hq_dyn = np.array([hq[j] for j in self.human_dyn])
        v['q'][0, nr:], v['qt'][0, nr:] = hq_dyn, hq_dyn
        v['tremor'][0], v['tremor_target'][0] = tremors, hq_dyn
        v['frozen'][0] = 0 if (impairment == 'tremor' or b.is_coop) else (((1 << b.nhdof) - 1) << nr)
        v['limit_scale'][0] = limit_scale
        v['base'][0, :3], v['base'][0, 3:] = self.base_pos, self.base_quat
        tp, tq = kin.tool_pose(self.base_pos, self.base_quat, q)                   # tool.py:49-62
        free = v['free'][0]
        free[:] = 0
        free[:, 6] = 1.0
        free[b.h['TOOL_BODY'], :3], free[b.h['TOOL_BODY'], 3:7] = tp, tq
        water_row[0] = (self.x0 + tp).astype(np.float32)                           # drinking.py:163-167: the grid is added to cup_pos, world axes
        water_row[1] = 0
        v['plane_friction'][0] = plane_friction
        v['gender'][0] = 0 if gender == 'male' else 1
        v['target'][0] = target
        v['iteration'][0], v['task_success'][0] = 0, 0
        v['total_food'][0] = len(self.x0)                                          # total_water_count (drinking.py:171)
        task = v['task'][0]
        task[:] = 0
        task[L.DK['ALIVE']:L.DK['ALIVE'] + 2] = -1                                 # self.waters / self.waters_active: all 64 (drinking.py:168-172)
        task[L.DK['ACTIVE']:L.DK['ACTIVE'] + 2] = -1
        v['rng'][0, 0] = (env_seed * 2654435761 + 12345) & 0x7FFFFFFF
        v['rng'][0, 1] = (env_seed ^ 0x5bd1e995) & 0x7FFFFFFF
        if info is not None:
            info.update(gender=gender, impairment=impairment, limit_scale=limit_scale, strength=strength, tremors=tremors, ik_ok=ok, ik_restarts=restarts,
                        ik_pos_err=best_d, target_ee_pos=target_ee_pos, cup_pos=tp)
        return state_row


def make_states(blob, n, seed=1001, impairment='random', **kw):
    """n independent pre-settle states and their water buffers (float32 [n, 2, 64, 3]); env i uses RandomState(seed + i)"""
    rs = DrinkingReset(blob)
    st = blob.new_state(n)
    water = np.zeros((n, 2, len(rs.x0), 3), dtype=np.float32)
    infos = [{} for _ in range(n)]
    for i in range(n):
        rs.sample(np.random.RandomState(seed + i), st[i:i + 1], water[i], env_seed=seed + i, impairment=impairment, info=infos[i], **kw)
    return st, water, infos
