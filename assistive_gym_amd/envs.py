"""gym.Env surface of the reference for the built hot paths (FeedingJaco-v1, BedBathingSawyer-v1 and their co-op flavours).

Mirrors, by name and behaviour, what assistive_gym/learn.py and env_viewer.py touch
(SURVEY 8b): classes ``FeedingJacoEnv`` (assistive_gym/envs/feeding_envs.py:29-31) and ``BedBathingSawyerEnv``
(assistive_gym/envs/bed_bathing_envs.py:23-25) with
``reset() -> obs``, ``step(a) -> (obs, reward, done, info)`` (feeding.py:12-43), ``seed``
(env.py:78-80), ``set_seed``, ``disconnect``, ``render`` (no-op: rendering is out of scope),
``action_space`` / ``observation_space`` (+ ``_robot`` / ``_human`` variants, env.py:42-49),
``action_robot_len`` ..., and the ``info`` keys of feeding.py:36.

Each scalar env is slot 0 of its own 1-environment stepper handle; the batched form is
assistive_gym_amd.vec_env.FeedingJacoVecEnv.  Physics always runs in libagx on the GPU: there is no
CPU fallback, constructing the env without a GPU raises.
"""
import numpy as np

from .blob import ModelBlob

try:                                     # gym is optional here (not installed in the build image)
    import gym
    from gym import spaces
    from gym.utils import seeding
    _Base = gym.Env
except Exception:                        # minimal stand-ins with the attributes the callers read
    gym = None

    class _Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low, self.high, self.dtype = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype), dtype
            self.shape = self.low.shape
            self._rng = np.random.RandomState()

        def seed(self, seed=None):
            self._rng = np.random.RandomState(seed)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

    class spaces:                        # noqa: N801
        Box = _Box

    class seeding:                       # noqa: N801
        @staticmethod
        def np_random(seed=None):
            if seed is None:
                seed = np.random.SeedSequence().entropy % (2 ** 31)
            return np.random.RandomState(int(seed) % (2 ** 32)), seed

    class _Base:
        metadata = {}

SETTLE_STEPS = 25       # feeding.py:178-179


def _box(n, bound):
    v = np.ones(n, dtype=np.float32) * bound
    return spaces.Box(low=-v, high=v, dtype=np.float32)


class _Agent:
    """The attributes of the reference's Robot / Human / Tool objects that learn.py, env_viewer.py and user scripts read
    (controllable_joint_indices, agent.py:21; Robot tables, robot.py:6-39).  Physics calls on them do not exist here."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class AssistiveEnv(_Base):
    """Common part of the scalar envs (AssistiveEnv, assistive_gym/envs/env.py:21-67)."""
    coop = False
    model = None            # blob name
    task = None

    def __init__(self, device=0):
        self.time_step, self.frame_skip = 0.02, 5                                    # env.py:21
        base = ModelBlob.load(self.model)
        self.blob = base.coop() if self.coop else base
        self.device = device
        self.action_robot_len, self.obs_robot_len = base.act_dim, base.obs_dim       # env.py:40-44, feeding.py:10
        self.action_human_len = self.blob.act_dim - base.act_dim                     # controllable human joints (feeding_envs.py:11)
        self.obs_human_len = self.blob.obs_dim - base.obs_dim                        # feeding.py:10: 23 in co-op
        self.action_space = _box(self.action_robot_len + self.action_human_len, 1.0)               # env.py:42
        self.observation_space = _box(self.obs_robot_len + self.obs_human_len, 1000000000.0)       # env.py:45
        self.action_space_robot, self.observation_space_robot = _box(self.action_robot_len, 1.0), _box(self.obs_robot_len, 1000000000.0)
        self.action_space_human, self.observation_space_human = _box(self.action_human_len, 1.0), _box(self.obs_human_len, 1000000000.0)
        self.gui = False
        self.iteration = 0
        self.task_success = 0
        self.total_force_on_human = 0.0
        self._stepper = None
        arm_pb = [base.robot_i(d, 'PB_INDEX') for d in sorted((d for d in range(base.nrobot) if base.robot_i(d, 'ACT') >= 0 and base.robot_i(d, 'ACT_SRC') == 0), key=lambda d: base.robot_i(d, 'ACT'))]
        if base.act_dim_robot == 2 * len(arm_pb):                                     # a single-arm robot with robot_arm = 'both' (robot.py:16)
            arm_pb = arm_pb + arm_pb
        hum_pb = [self.blob.robot_i(d, 'PB_INDEX') for d in range(base.nrobot, base.ndof)] if self.coop else []
        mobile = base.h['BASE_LINK'] > 0                                              # robot.py:11: 'wheel' in controllable_joints
        d0 = next(d for d in range(base.nrobot) if base.robot_i(d, 'ACT') >= 0)
        self.robot = _Agent(controllable_joint_indices=arm_pb, mobile=mobile, motor_gains=base.robot_f(d0, 'KP'), motor_forces=base.robot_f(d0, 'MAXF'),
                            wheel_joint_indices=[base.robot_i(d, 'PB_INDEX') for d in range(base.nrobot) if base.robot_i(d, 'OBS_SKIP') and base.robot_i(d, 'ACT_SRC') == 0 and base.robot_i(d, 'ACT') >= 0])
        self.human = _Agent(controllable_joint_indices=hum_pb, controllable=self.coop)
        self.tool = _Agent()
        self.camera_width, self.camera_height, self.view_matrix, self.projection_matrix = None, None, None, None
        self.seed(1001)                                                               # env.py:21,30

    # ---- gym API ------------------------------------------------------------------------------
    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def set_seed(self, seed=1000):
        self.np_random.seed(seed)

    def _ensure_stepper(self):
        if self._stepper is None:
            from .libagx import Stepper          # raises without the HIP library / a GPU
            self._stepper = Stepper(self.blob, 1, self.device)
        return self._stepper

    def _draw_seed(self):
        """62 bits from the gym-seeded generator: the key of the device-side reset generator's counter RNG"""
        r = self.np_random
        draw = r.integers if hasattr(r, 'integers') else r.randint
        return (int(draw(0, 2 ** 31 - 1)) << 31) | int(draw(0, 2 ** 31 - 1))

    def reset(self):
        raise NotImplementedError('Implement reset')                                              # env.py:69-70

    def _split_obs(self, obs):
        if not self.coop:
            return obs
        return {'robot': obs[:self.obs_robot_len], 'human': obs[self.obs_robot_len:]}              # feeding.py:110-111

    def step(self, action):
        if self.coop and isinstance(action, dict):                                                 # feeding.py:13-14
            action = np.concatenate([action['robot'], action['human']])
        action = np.asarray(action, dtype=np.float32)
        n_act = self.action_robot_len + self.action_human_len
        if action.shape != (n_act,):
            # the reference prints and exit()s (env.py:198-200); a library must not kill the process
            raise ValueError('Received agent actions of length %d does not match expected action length of %d' % (action.size, n_act))
        obs, rew, done, info = self._ensure_stepper().step_host(action[None])
        self.iteration += 1
        self.total_force_on_human = float(info[0, 0])
        out_info = {'total_force_on_human': float(info[0, 0]), 'task_success': int(info[0, 1]),
                    'action_robot_len': self.action_robot_len, 'action_human_len': self.action_human_len,
                    'obs_robot_len': self.obs_robot_len, 'obs_human_len': self.obs_human_len}      # feeding.py:36
        o, r, d = self._split_obs(obs[0].astype(np.float64)), float(rew[0]), bool(done[0])
        if not self.coop:
            return o, r, d, out_info
        # co-optimisation: per-agent dictionaries (feeding.py:41-43)
        return o, {'robot': r, 'human': r}, {'robot': d, 'human': d, '__all__': d}, {'robot': out_info, 'human': out_info}

    def render(self, mode='human'):
        return None          # GUI / EGL rendering is outside the hot path

    def setup_camera(self, camera_eye=(0.5, -0.75, 1.5), camera_target=(-0.2, 0, 0.75), fov=60, camera_width=1920 // 4, camera_height=1080 // 4):
        """env.py:336-340: remembered only; there is no renderer behind it"""
        self.camera_width, self.camera_height = camera_width, camera_height
        self.view_matrix, self.projection_matrix = (tuple(camera_eye), tuple(camera_target)), fov

    def get_camera_image_depth(self, *a, **kw):
        assert self.view_matrix is not None, 'You must call env.setup_camera() or env.setup_camera_rpy() before getting a camera image'   # env.py:355
        # rendering is out of scope: a blank frame of the requested size keeps learn.render_policy's loop (learn.py:96-131) running
        return np.zeros((self.camera_height, self.camera_width, 4), dtype=np.uint8), np.ones((self.camera_height, self.camera_width), dtype=np.float32)

    def disconnect(self):
        if self._stepper is not None:
            self._stepper.close()
            self._stepper = None

    close = disconnect

    # ---- state injection (no reference equivalent; parity protocol of SURVEY 8c) ----------------
    def get_state(self):
        return self._ensure_stepper().get_state()[0]

    def set_state(self, state):
        self._ensure_stepper().set_state(np.asarray(state, dtype=np.float32).reshape(1, -1))


class FeedingJacoEnv(AssistiveEnv):
    """FeedingJaco-v1: Jaco arm on a wheelchair feeds a static (non-cooperating) human."""
    model, task = 'feeding_jaco', 'feeding'

    def reset(self):
        """FeedingEnv.reset (feeding.py:114-182): sampled on the device (agx_sample_reset: human, IK with random
        restarts, tool / bowl / food), settled for 25 substeps, observed."""
        st = self._ensure_stepper()
        self.reset_seed = self._draw_seed()
        st.sample_reset(self.reset_seed, impairment='random')
        st.settle(SETTLE_STEPS)
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))


class FeedingJacoHumanEnv(FeedingJacoEnv):
    """FeedingJacoHuman-v1 (feeding_envs.py:64-67): the human's head joints are controllable; actions,
    observations, rewards, dones and infos are per-agent dictionaries as RLlib's MultiAgentEnv expects."""
    coop = True


class FeedingSawyerEnv(FeedingJacoEnv):
    """FeedingSawyer-v1 (feeding_envs.py:25-27): a free-standing robot; its base pose comes from the base pose search on the host
    (host/reset.py + host/reset_bed.py) with the device's collision pass rejecting colliding placements, then the 25 settle steps."""
    model = 'feeding_sawyer'

    def reset(self):
        from .host.reset import make_states
        from .host.reset_bed import DeviceCollisionChecker
        st = self._ensure_stepper()
        if not hasattr(self, '_checker'):
            self._checker = DeviceCollisionChecker(self.blob, 1, self.device)
        self.reset_seed = self._draw_seed()
        rec, _ = make_states(self.blob, 1, seed=self.reset_seed % (2 ** 31), checker=self._checker)
        st.set_state(rec)
        st.settle(SETTLE_STEPS)
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))


class FeedingSawyerHumanEnv(FeedingSawyerEnv):
    """FeedingSawyerHuman-v1 (feeding_envs.py:51-54)"""
    coop = True


class FeedingBaxterEnv(FeedingSawyerEnv):
    """FeedingBaxter-v1 (feeding_envs.py:21-23): Baxter's right arm"""
    model = 'feeding_baxter'


class FeedingBaxterHumanEnv(FeedingBaxterEnv):
    """FeedingBaxterHuman-v1 (feeding_envs.py:46-49)"""
    coop = True


class FeedingPandaEnv(FeedingJacoEnv):
    """FeedingPanda-v1 (feeding_envs.py:35-37): the same task with the wheelchair-mounted Franka Panda (agents/panda.py)."""
    model = 'feeding_panda'


class FeedingPandaHumanEnv(FeedingPandaEnv):
    """FeedingPandaHuman-v1 (feeding_envs.py:69-73)"""
    coop = True


class DrinkingJacoEnv(AssistiveEnv):
    """DrinkingJaco-v1 (drinking_envs.py:29-31): the wheelchair-mounted Jaco tilts a cup with 64 water particles at the person's mouth.
    The water lives next to the state record on the device (float32 [2, 64, 3]: positions, velocities)."""
    model, task = 'drinking_jaco', 'drinking'

    def reset(self):
        """DrinkingEnv.reset (drinking.py:122-181) on the device: human, target, robot start pose (IK restarts / base pose search / placement
        draws with collision rejection), cup, the 4 x 4 x 4 water grid above it, then the 50 steps in which the water drops into the cup"""
        st = self._ensure_stepper()
        self.reset_seed = self._draw_seed()
        st.sample_reset(self.reset_seed, impairment='random')
        st.settle(50)                                                                              # drinking.py:176-177
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))

    # the water lives outside the state record: a state is the pair (record, water[2][64][3])
    def get_state(self):
        st = self._ensure_stepper()
        return st.get_state()[0], st.get_cloth()[0]

    def set_state(self, state):
        if not (isinstance(state, (tuple, list)) and len(state) == 2):
            raise ValueError('a drinking state is the pair (state record, water): pass what get_state() returned')
        st = self._ensure_stepper()
        st.set_state(np.asarray(state[0], dtype=np.float32).reshape(1, -1))
        st.set_cloth(np.ascontiguousarray(state[1], dtype=np.float32)[None])


class DrinkingJacoHumanEnv(DrinkingJacoEnv):
    """DrinkingJacoHuman-v1 (drinking_envs.py:56-59): the human's head joints are controllable too"""
    coop = True


class BedBathingSawyerEnv(AssistiveEnv):
    """BedBathingSawyer-v1 (bed_bathing_envs.py:23-25): Sawyer wipes the right arm of a human lying on a bed."""
    model, task = 'bed_bathing_sawyer', 'bed_bathing'

    def reset(self):
        """BedBathingEnv.reset (bed_bathing.py:112-171): the draws, the TOC base pose search and the IK restated on the host
        (host/reset_bed.py) around the rag-doll settle of the human, which runs on the device (bed_settle model); the result is
        injected into the stepper."""
        from .host.reset_bed import make_states, RagdollSettler, DeviceCollisionChecker
        st = self._ensure_stepper()
        if not hasattr(self, '_settler'):
            self._settler, self._checker = RagdollSettler(1, self.device), DeviceCollisionChecker(self.blob, 1, self.device)
        self.reset_seed = self._draw_seed()
        rec, _ = make_states(self.blob, 1, seed=self.reset_seed % (2 ** 31), settler=self._settler, checker=self._checker)
        st.set_state(rec)
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))


class BedBathingSawyerHumanEnv(BedBathingSawyerEnv):
    """BedBathingSawyerHuman-v1 (bed_bathing_envs.py:57-61): the human's right arm (10 joints) is controllable; the pose-dependent
    arm limits (human.py:134-152, the Keras classifier) run after every substep; per-agent dictionaries as in the reference."""
    coop = True


class ScratchItchPR2Env(AssistiveEnv):
    """ScratchItchPR2-v1 (scratch_itch_envs.py:17-19): the PR2's left arm scratches a target on the seated human's right arm."""
    model, task = 'scratch_itch_pr2', 'scratch_itch'

    def reset(self):
        """ScratchItchEnv.reset (scratch_itch.py:93-132), restated on the host (host/reset_scratch.py); its result is injected."""
        from .host.reset_scratch import make_states
        from .host.reset_bed import DeviceCollisionChecker
        st = self._ensure_stepper()
        if not hasattr(self, '_checker'):
            self._checker = DeviceCollisionChecker(self.blob, 1, self.device)
        self.reset_seed = self._draw_seed()
        rec, _ = make_states(self.blob, 1, seed=self.reset_seed % (2 ** 31), checker=self._checker)
        st.set_state(rec)
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))


class ScratchItchPR2HumanEnv(ScratchItchPR2Env):
    """ScratchItchPR2Human-v1 (scratch_itch_envs.py:41-44; BASELINE config 4): the human's right arm (10 joints) is controllable,
    the pose-dependent arm limits run after every substep; actions {'robot': a[7], 'human': a[10]}, observations 30 + 34."""
    coop = True


class ScratchItchJacoEnv(ScratchItchPR2Env):
    """ScratchItchJaco-v1 (scratch_itch_envs.py:29-31; the default environment of the reference's env_viewer.py / learn.py): the
    wheelchair-mounted Jaco holds the scratcher.  reset() is sampled on the device (agx_sample_reset: human, IK with random restarts and
    collision rejection, tool, target on the arm), as FeedingJacoEnv's."""
    model = 'scratch_itch_jaco'

    def reset(self):
        st = self._ensure_stepper()
        self.reset_seed = self._draw_seed()
        st.sample_reset(self.reset_seed, impairment='random')
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))


class ScratchItchJacoHumanEnv(ScratchItchJacoEnv):
    coop = True


class ScratchItchPandaEnv(ScratchItchJacoEnv):
    """ScratchItchPanda-v1 (scratch_itch_envs.py:37-39): as ScratchItchJaco, with the wheelchair-mounted Panda"""
    model = 'scratch_itch_panda'


class ScratchItchPandaHumanEnv(ScratchItchPandaEnv):
    coop = True


class ScratchItchSawyerEnv(ScratchItchPR2Env):
    """ScratchItchSawyer-v1 (scratch_itch_envs.py:25-27)"""
    model = 'scratch_itch_sawyer'


class ScratchItchSawyerHumanEnv(ScratchItchSawyerEnv):
    coop = True


class DressingBaxterEnv(AssistiveEnv):
    """DressingBaxter-v1 (dressing_envs.py:19-21; BASELINE config 5): Baxter's left arm pulls the sleeve of a hospital gown over the left
    arm of the seated human.  The garment (3,966 nodes) lives next to the state record on the device."""
    model, task = 'dressing_baxter', 'dressing'

    def reset(self):
        """DressingEnv.reset (dressing.py:112-198): the draws, the TOC base pose search and the IK restated on the host
        (host/reset_dressing.py); the garment is loaded relative to the end effector and settles for 50 steps on the device."""
        from .host.reset_dressing import make_states, ClothSettler
        from .host.reset_bed import DeviceCollisionChecker
        st = self._ensure_stepper()
        if not hasattr(self, '_settler'):
            self._settler, self._checker = ClothSettler(self.blob, 1, self.device), DeviceCollisionChecker(self.blob, 1, self.device)
        self.reset_seed = self._draw_seed()
        rec, cloth, _ = make_states(self.blob, 1, seed=self.reset_seed % (2 ** 31), settler=self._settler, checker=self._checker)
        st.set_state(rec)
        st.set_cloth(cloth)
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))

    # the garment lives outside the state record: a state is the pair (record, garment[2][NN][3])
    def get_state(self):
        st = self._ensure_stepper()
        return st.get_state()[0], st.get_cloth()[0]

    def set_state(self, state):
        if not (isinstance(state, (tuple, list)) and len(state) == 2):
            raise ValueError('a dressing state is the pair (state record, garment): pass what get_state() returned')
        st = self._ensure_stepper()
        st.set_state(np.asarray(state[0], dtype=np.float32).reshape(1, -1))
        st.set_cloth(np.ascontiguousarray(state[1], dtype=np.float32)[None])


class DressingBaxterHumanEnv(DressingBaxterEnv):
    """DressingBaxterHuman-v1 (dressing_envs.py:51-55): the human's left arm (10 joints) is controllable, the pose-dependent arm limits
    run after every stepSimulation; actions {'robot': a[7], 'human': a[10]}, observations 24 + 28."""
    coop = True


class ArmManipulationSawyerEnv(AssistiveEnv):
    """ArmManipulationSawyer-v1 (arm_manipulation_envs.py:23-25): Sawyer lifts the limp right arm of a human lying on a bed back onto
    the body with a scooper.  robot_arm = 'both' on a single-arm robot: 14 actions (the arm joints twice, robot.py:16)."""
    model, task = 'arm_manipulation_sawyer', 'arm_manipulation'

    def reset(self):
        """ArmManipulationEnv.reset (arm_manipulation.py:110-182): host/reset_arm.py around the two settles on the device."""
        from .host.reset_arm import make_states, ArmFallSettler
        from .host.reset_bed import RagdollSettler, DeviceCollisionChecker
        st = self._ensure_stepper()
        if not hasattr(self, '_settler'):
            self._settler, self._arm_settler = RagdollSettler(1, self.device), ArmFallSettler(self.blob, 1, self.device)
            self._checker = DeviceCollisionChecker(self.blob, 1, self.device)
        self.reset_seed = self._draw_seed()
        rec, _ = make_states(self.blob, 1, seed=self.reset_seed % (2 ** 31), settler=self._settler, arm_settler=self._arm_settler, checker=self._checker)
        st.set_state(rec)
        self.iteration, self.task_success = 0, 0
        return self._split_obs(st.observe_host()[0].astype(np.float64))


class ArmManipulationSawyerHumanEnv(ArmManipulationSawyerEnv):
    """ArmManipulationSawyerHuman-v1 (arm_manipulation_envs.py:57-61): the human's right arm (10 joints) is controllable;
    actions {'robot': a[14], 'human': a[10]}, observations 45 + 42."""
    coop = True


ENV_IDS = {'FeedingSawyer-v1': FeedingSawyerEnv, 'FeedingSawyerHuman-v1': FeedingSawyerHumanEnv, 'FeedingBaxter-v1': FeedingBaxterEnv, 'FeedingBaxterHuman-v1': FeedingBaxterHumanEnv,
           'ScratchItchJaco-v1': ScratchItchJacoEnv, 'ScratchItchJacoHuman-v1': ScratchItchJacoHumanEnv, 'ScratchItchPanda-v1': ScratchItchPandaEnv,
           'ScratchItchPandaHuman-v1': ScratchItchPandaHumanEnv, 'ScratchItchSawyer-v1': ScratchItchSawyerEnv, 'ScratchItchSawyerHuman-v1': ScratchItchSawyerHumanEnv,
           'FeedingPanda-v1': FeedingPandaEnv, 'FeedingPandaHuman-v1': FeedingPandaHumanEnv, 'ArmManipulationSawyer-v1': ArmManipulationSawyerEnv, 'ArmManipulationSawyerHuman-v1': ArmManipulationSawyerHumanEnv, 'DressingBaxter-v1': DressingBaxterEnv, 'DressingBaxterHuman-v1': DressingBaxterHumanEnv, 'ScratchItchPR2-v1': ScratchItchPR2Env, 'ScratchItchPR2Human-v1': ScratchItchPR2HumanEnv, 'FeedingJaco-v1': FeedingJacoEnv, 'FeedingJacoHuman-v1': FeedingJacoHumanEnv, 'BedBathingSawyer-v1': BedBathingSawyerEnv,
           'BedBathingSawyerHuman-v1': BedBathingSawyerHumanEnv, 'DrinkingJaco-v1': DrinkingJacoEnv, 'DrinkingJacoHuman-v1': DrinkingJacoHumanEnv}

# ---- further robots of a task: the same env code with another model blob (model/compiler.py ROBOT_BASE / ROBOT_TASK) -----------------------
def _robot_flavours(base_cls, task_name, robots, ref):
    for robot, model in robots:
        cls = type('%s%sEnv' % (task_name, robot), (base_cls,), {'model': model, '__doc__': '%s%s-v1 (%s): %s with the %s' % (task_name, robot, ref, base_cls.__name__, robot)})
        coop_cls = type('%s%sHumanEnv' % (task_name, robot), (cls,), {'coop': True, '__doc__': '%s%sHuman-v1 (%s): the human is controllable too' % (task_name, robot, ref)})
        globals()[cls.__name__], globals()[coop_cls.__name__] = cls, coop_cls
        ENV_IDS['%s%s-v1' % (task_name, robot)] = cls
        ENV_IDS['%s%sHuman-v1' % (task_name, robot)] = coop_cls


_robot_flavours(BedBathingSawyerEnv, 'BedBathing', [('Jaco', 'bed_bathing_jaco'), ('Panda', 'bed_bathing_panda'), ('PR2', 'bed_bathing_pr2'), ('Baxter', 'bed_bathing_baxter'), ('Stretch', 'bed_bathing_stretch')],
                'bed_bathing_envs.py:15-37,45-79')
_robot_flavours(FeedingSawyerEnv, 'Feeding', [('PR2', 'feeding_pr2'), ('Stretch', 'feeding_stretch')], 'feeding_envs.py:17-19,33-35,41-44,60-63')
_robot_flavours(DressingBaxterEnv, 'Dressing', [('Sawyer', 'dressing_sawyer'), ('Jaco', 'dressing_jaco'), ('Panda', 'dressing_panda'), ('PR2', 'dressing_pr2'), ('Stretch', 'dressing_stretch')], 'dressing_envs.py:23-37,56-79')
_robot_flavours(ArmManipulationSawyerEnv, 'ArmManipulation', [('Jaco', 'arm_manipulation_jaco'), ('Panda', 'arm_manipulation_panda'), ('PR2', 'arm_manipulation_pr2'),
                                                              ('Baxter', 'arm_manipulation_baxter')], 'arm_manipulation_envs.py:15-37,41-79')
_robot_flavours(ScratchItchPR2Env, 'ScratchItch', [('Baxter', 'scratch_itch_baxter'), ('Stretch', 'scratch_itch_stretch')], 'scratch_itch_envs.py:21-23,31-33,46-50,58-62')
_robot_flavours(DrinkingJacoEnv, 'Drinking', [('PR2', 'drinking_pr2'), ('Baxter', 'drinking_baxter'), ('Sawyer', 'drinking_sawyer'), ('Stretch', 'drinking_stretch'), ('Panda', 'drinking_panda')],
                'drinking_envs.py:15-27,33-55,61-67')



def make(env_id):
    """`gym.make('assistive_gym:FeedingJaco-v1')` equivalent, with the TimeLimit of 200 steps folded
    into the env itself (done = iteration >= 200, feeding.py:37; assistive_gym/__init__.py:11)."""
    name = env_id.split(':')[-1]
    if name not in ENV_IDS:
        raise KeyError('%s is not built yet (hot-path scope: %s)' % (env_id, sorted(ENV_IDS)))
    return ENV_IDS[name]()


if gym is not None:       # same ids the reference registers (assistive_gym/__init__.py:6-12)
    try:
        from gym.envs.registration import register
        register(id='FeedingJaco-v1', entry_point='assistive_gym_amd.envs:FeedingJacoEnv', max_episode_steps=200)
        register(id='BedBathingSawyer-v1', entry_point='assistive_gym_amd.envs:BedBathingSawyerEnv', max_episode_steps=200)
        register(id='ScratchItchPR2-v1', entry_point='assistive_gym_amd.envs:ScratchItchPR2Env', max_episode_steps=200)
    except Exception:
        pass
