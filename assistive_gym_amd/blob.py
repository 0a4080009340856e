"""Model blob reader + per-environment state record packing (layout: include/agx_blob.h)."""
import json
import os

import numpy as np

from .model import compiler as L   # layout constant tables only (H, P, R, F, C, G, T, E)

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


class ModelBlob:
    def __init__(self, words, meta=None):
        self.words = np.ascontiguousarray(words, dtype=np.uint32)
        self.f = self.words.view(np.float32)
        self.i = self.words.view(np.int32)
        assert self.i[L.H['MAGIC']] == L.MAGIC and self.i[L.H['VERSION']] == L.VERSION, 'bad model blob'
        self.meta = meta or {}
        self.h = {k: int(self.i[v]) for k, v in L.H.items() if k != 'COUNT'}
        self.ndof, self.nfree, self.nhuman = self.h['NDOF'], self.h['NFREE'], self.h['NHUMAN']
        self.nfood, self.act_dim, self.obs_dim = self.h['NFOOD'], self.h['ACT_DIM'], self.h['OBS_DIM']
        self.state_words = self.h['STATE_WORDS']
        self.nrobot, self.nhdof = self.h['NROBOT'], self.h['NHDOF']
        self.task_kind = self.h['TASK_KIND']

    @classmethod
    def load(cls, name='feeding_jaco'):
        path = os.path.join(DATA_DIR, name + '.agxblob')
        meta_path = os.path.join(DATA_DIR, name + '.meta.json')
        meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
        return cls(np.fromfile(path, dtype=np.uint32), meta)

    # ---- sections --------------------------------------------------------------------------
    @property
    def has_reset_generator(self):
        """the blob carries a reset section the device-side reset generator (csrc/agx_reset.h, agx_sample_reset / agx_reset) can sample from:
        every feeding, scratch-itch, dressing and bed-bathing scene (wheelchair-mounted arm: IK restarts; free-standing robot: base pose
        search; robot on wheels: placement draws; bed bathing: with the rag-doll model attached, agx_attach_settle_model), the rag-doll
        model itself (its drop record) and the arm-manipulation scenes (with their fall model -- fall_model() -- and the rag-doll model
        behind it; PR2 / Baxter: one base pose for two arm chains)"""
        from .model import compiler as L
        x0 = int(self.i[L.H['OFF_RESET']])
        words = int(self.i[L.H['OFF_TARGETS']]) - x0 if 'OFF_TARGETS' in L.H else 0
        return words > L.X_['COUNT'] and int(self.i[x0 + L.X_['NARM']]) > 0

    def param(self, key):
        return float(self.f[self.h['OFF_PARAMS'] + L.P[key]])

    def set_param(self, key, value):
        """Returns a copy of the blob with one PARAMS entry changed (tests, ablations)."""
        w = self.words.copy()
        w.view(np.float32)[self.h['OFF_PARAMS'] + L.P[key]] = value
        return ModelBlob(w, self.meta)

    def fall_model(self):
        """The model the arm of ArmManipulationEnv.reset falls in (arm_manipulation.py:139-146): this blob at the reset's gravity of -1 on the
        human, its sampler switched to the record the arm falls from (AGX_X_FLAGS bit 8).  A second handle on it is attached to the task's
        handle, the rag-doll handle to that one (agx_attach_settle_model twice)."""
        assert self.task_kind == L.TASK_ARM_MANIPULATION and self.has_reset_generator
        w = self.words.copy()
        w.view(np.float32)[self.h['OFF_PARAMS'] + L.P['HUMAN_GRAVITY_Z']] = -1.0
        w.view(np.int32)[int(self.i[L.H['OFF_RESET']]) + L.X_['FLAGS']] |= 256
        return ModelBlob(w, self.meta)

    def with_drop_base(self, pos):
        """The rag-doll model (bed_settle) dropping the human from another spot: ArmManipulationEnv.reset drops it from [-0.25, 0.2, 0.95]
        (arm_manipulation.py:123), BedBathingEnv.reset from [-0.15, 0.2, 0.95] (bed_bathing.py:121, the compiled value)."""
        x0 = int(self.i[L.H['OFF_RESET']])
        assert int(self.i[x0 + L.X_['FLAGS']]) & 32
        w = self.words.copy()
        for key in ('HBASE_M', 'HBASE_F'):
            w.view(np.float32)[x0 + L.X_[key]:x0 + L.X_[key] + 3] = pos
        return ModelBlob(w, self.meta)

    def coop(self):
        """Returns the co-op flavour of this blob (<Task><Robot>HumanEnv, feeding_envs.py:64-67): the human's
        controllable joints (head, ACT indices after the robot's) take actions and the observation is
        followed by the human's (feeding.py:102-111)."""
        w = self.words.copy()
        wi = w.view(np.int32)
        n_h = sum(1 for d in range(self.nrobot, self.ndof) if self.robot_i(d, 'ACT') >= 0)
        wi[L.H['ACT_DIM']] = self.act_dim_robot + n_h
        # obs_human_len: 19 + joints in feeding.py:10, 18 + joints in bed_bathing.py:10
        wi[L.H['OBS_DIM']] = self.obs_dim_robot + {L.TASK_BED_BATHING: 18, L.TASK_SCRATCH_ITCH: 24, L.TASK_DRESSING: 18, L.TASK_ARM_MANIPULATION: 32}.get(self.task_kind, 19) + n_h    # scratch_itch.py:8, dressing.py:9, arm_manipulation.py:11
        wi[self.h['OFF_TASK'] + L.T['COOP']] = 1
        # pose-dependent arm limits (human.py:134-152) run when a shoulder joint is controllable (human.py:136-137)
        if self.h['OFF_MLP'] and any(self.robot_i(d, 'ACT') >= 0 for d in self.task_i_n('ARM_LIMIT_DOF', 1)):
            wi[self.h['OFF_TASK'] + L.T['ARM_LIMIT_ON']] = 1
        return ModelBlob(w, self.meta)

    @property
    def is_coop(self):
        return self.task_i('COOP') == 1

    @property
    def act_dim_robot(self):
        # a single-arm robot driven with robot_arm = 'both' lists its arm joints twice (robot.py:16): DUP_ACT more actions
        # (joints that share another joint's action -- Robot.action_duplication, AGX_R_ACT_SRC -- add none)
        return sum(1 for d in range(self.nrobot) if self.robot_i(d, 'ACT') >= 0 and self.robot_i(d, 'ACT_SRC') == 0) + (self.task_i('DUP_ACT') if self.task_kind == L.TASK_ARM_MANIPULATION else 0)

    @property
    def obs_dim_robot(self):
        skipped = sum(1 for d in range(self.nrobot) if self.robot_i(d, 'ACT') >= 0 and self.robot_i(d, 'ACT_SRC') == 0 and self.robot_i(d, 'OBS_SKIP'))     # a mobile robot's wheels (feeding.py:90-92)
        return {L.TASK_BED_BATHING: 17, L.TASK_SCRATCH_ITCH: 23, L.TASK_DRESSING: 17, L.TASK_ARM_MANIPULATION: 31}.get(self.task_kind, 18) + self.act_dim_robot - skipped       # bed_bathing.py:10 / scratch_itch.py:8 / dressing.py:9 / feeding.py:10

    def rec(self, d, gender=0):
        """link record index of DoF d (human DoFs have one record per gender)"""
        return d if d < self.nrobot else d + gender * self.nhdof

    def robot_f(self, d, key, n=1, gender=0):
        o = self.h['OFF_ROBOT'] + self.rec(d, gender) * L.R['STRIDE'] + L.R[key]
        return self.f[o:o + n].astype(np.float64) if n > 1 else float(self.f[o])

    def robot_i(self, d, key, gender=0):
        return int(self.i[self.h['OFF_ROBOT'] + self.rec(d, gender) * L.R['STRIDE'] + L.R[key]])

    def free_f(self, b, key, n=1):
        o = self.h['OFF_FREE'] + b * L.F['STRIDE'] + L.F[key]
        return self.f[o:o + n].astype(np.float64) if n > 1 else float(self.f[o])

    def task_f(self, key, n=1):
        o = self.h['OFF_TASK'] + L.T[key]
        return self.f[o:o + n].astype(np.float64) if n > 1 else float(self.f[o])

    def task_i(self, key):
        return int(self.i[self.h['OFF_TASK'] + L.T[key]])

    def task_i_n(self, key, n):
        o = self.h['OFF_TASK'] + L.T[key]
        return [int(x) for x in self.i[o:o + n]]

    def collider(self, c):
        o = self.h['OFF_COLL'] + c * L.C['STRIDE']
        nv, vo = int(self.i[o + L.C['NVERT']]), int(self.i[o + L.C['VOFF']])
        v0 = self.h['OFF_VERT'] + 3 * vo
        return dict(body=int(self.i[o + L.C['BODY']]), radius=float(self.f[o + L.C['RADIUS']]),
                    friction=float(self.f[o + L.C['FRICTION']]), tag=int(self.i[o + L.C['TAG']]), link=int(self.i[o + L.C['LINK']]),
                    verts=self.f[v0:v0 + 3 * nv].reshape(nv, 3).astype(np.float64))

    # ---- state records ------------------------------------------------------------------------
    def new_state(self, n=1):
        return np.zeros((n, self.state_words), dtype=np.float32)

    def view(self, state):
        """Named views into a (n, state_words) float32 array (writes go through)."""
        h = self.h
        s = state.reshape(-1, self.state_words)
        si = s.view(np.int32)
        e = h['S_ENV']
        return dict(
            q=s[:, h['S_Q']:h['S_Q'] + self.ndof], qd=s[:, h['S_QD']:h['S_QD'] + self.ndof],
            qt=s[:, h['S_QT']:h['S_QT'] + self.ndof],
            free=s[:, h['S_FREE']:h['S_FREE'] + 13 * self.nfree].reshape(len(s), self.nfree, 13),
            base=s[:, h['S_BASE']:h['S_BASE'] + 7],
            human=s[:, h['S_HUMAN']:h['S_HUMAN'] + 7 * self.nhuman].reshape(-1, self.nhuman, 7),
            plane_friction=s[:, e + L.E['PLANE_FRICTION']], gender=si[:, e + L.E['GENDER']],
            target=s[:, e + L.E['TARGET']:e + L.E['TARGET'] + 3],
            food_alive=si[:, e + L.E['FOOD_ALIVE']], food_active=si[:, e + L.E['FOOD_ACTIVE']],
            iteration=si[:, e + L.E['ITERATION']], task_success=si[:, e + L.E['TASK_SUCCESS']],
            rng=si[:, e + L.E['RNG']:e + L.E['RNG'] + 2], total_food=si[:, e + L.E['TOTAL_FOOD']],
            frozen=si[:, e + L.E['FROZEN']], limit_scale=s[:, e + L.E['LIMIT_SCALE']],
            human_kp=s[:, e + L.E['HUMAN_KP']], human_maxf=s[:, e + L.E['HUMAN_MAXF']],
            task=si[:, h['S_TASK']:h['S_TASK'] + h['TASK_WORDS']],
            tremor=s[:, h['S_TREMOR']:h['S_TREMOR'] + self.nhdof],
            tremor_target=s[:, h['S_TREMOR'] + self.nhdof:h['S_TREMOR'] + 2 * self.nhdof])
