"""-m gpu: ArmManipulationSawyer-v1 and its co-op flavour on the HIP stepper (arm_manipulation kernel variant, through the C ABI)
against the CPU oracle on the same seeded inputs; single-step comparisons from the device's own states (the limp arm under full
gravity is in sustained contact with the mattress).  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

from test_arm_manipulation import _states, scooper_under_forearm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def am():
    from assistive_gym_amd import libagx
    from assistive_gym_amd.blob import ModelBlob
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    return ModelBlob.load('arm_manipulation_sawyer')


@pytest.fixture(scope='module')
def am_oracle(am):
    from oracle_lib import Oracle
    return Oracle(am)


@pytest.fixture(scope='module')
def device_fall(am):
    from assistive_gym_amd.host.reset_arm import ArmFallSettler
    return ArmFallSettler(am, 16)


def _check_step(blob, o, st, ref, act, worst):
    obs, rew, done, info = st.step_host(act)
    got = st.get_state()
    _check_step.tool_force = info[:, 3].copy()
    r = blob.obs_dim_robot
    for i in range(len(ref)):
        o_obs, o_rew, o_done, o_info = o.step(ref[i], act[i])
        assert info[i, 6] == o_info[6] and abs(info[i, 7] - o_info[7]) <= 2, (i, info[i], o_info)
        dev = np.abs(obs[i] - o_obs)
        forces = [r - 2, r - 1] + ([blob.obs_dim - 3, blob.obs_dim - 2, blob.obs_dim - 1] if blob.is_coop else [])
        for k in forces:
            assert dev[k] <= 1e-3 * max(1.0, abs(o_obs[k])), (i, k, obs[i, k], o_obs[k])
            dev[k] = 0
        # the reward carries 0.01 * pressure and 0.01 * forces: compare it at the forces' tolerance
        worst[i] = max(worst[i], float(dev.max()), abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)) * 0.1)
        assert info[i, 4] == o_info[4] and info[i, 1] == o_info[1] and bool(done[i]) == o_done
        for c in (0, 2, 3):
            assert abs(info[i, c] - o_info[c]) <= 1e-3 * max(1.0, abs(o_info[c])), (i, c, info[i], o_info)
        vg, vo = blob.view(got[i].reshape(1, -1)), blob.view(ref[i].reshape(1, -1))
        assert abs(vg['task'].view(np.float32)[0, 0] - vo['task'].view(np.float32)[0, 0]) < 1e-4
    return got


def test_variant(am):
    from assistive_gym_amd.libagx import Stepper
    st = Stepper(am, 2)
    assert st.variant() == 'arm_manipulation'
    st.close()


def test_arm_fall_on_the_device_matches_the_oracle(am, device_fall):
    """the second settle of the reset (100 stepSimulation calls at gravity -1): device vs oracle, from the same posed records"""
    from oracle_lib import Oracle
    import conditioning as C
    posed, _ = _states(am, 4, 7001)
    o = Oracle(device_fall.blob)
    fell = device_fall(posed, 100)
    fw = C.float_words(am)
    for i in range(4):
        ref = posed[i].copy()
        o.settle(ref, 100)
        # 100 FREE-RUNNING substeps of a limp arm falling onto the body (not a single step): the yardstick is the oracle's own run from
        # the same record moved by one float32 ulp per word (tests/conditioning.py)
        twin = posed[i].copy(); twin[fw] = C._perturb_f32(twin[fw], np.random.RandomState(i)); o.settle(twin, 100)
        q, qo, qt = am.view(fell[i:i + 1])['q'][0, 10:], am.view(ref.reshape(1, -1))['q'][0, 10:], am.view(twin.reshape(1, -1))['q'][0, 10:]
        spread = float(np.abs(qt - qo).max())
        print('arm fall env %d: device vs oracle %.3g rad, oracle vs its 1-ulp twin %.3g rad' % (i, np.abs(q - qo).max(), spread))
        assert np.abs(q - qo).max() < max(3e-4, C.K * spread), (i, q, qo, spread)
        assert np.abs(q - am.view(posed[i:i + 1])['q'][0, 10:]).max() > 0.05


def test_step_matches_oracle(am, am_oracle, device_fall):
    from assistive_gym_amd.libagx import Stepper
    a, _ = _states(am, 12, 7101, arm_settler=device_fall)
    w = [scooper_under_forearm(am, am_oracle, seed=7201 + k, depth=0.002 + 0.001 * k)[0] for k in range(4)]
    states = np.concatenate([a, np.array(w)])
    n = len(states)
    st = Stepper(am, n)
    st.set_state(states)
    worst = np.zeros(n)
    lifted = 0
    for k in range(4):
        act = np.random.RandomState(100 + k).uniform(-1, 1, (n, 14)).astype(np.float32)
        act[12:] *= 0.1
        ref = st.get_state()                                   # single-step comparison from the device's own state
        _check_step(am, am_oracle, st, ref, act, worst)
        lifted += int((_check_step.tool_force[12:] > 0).sum())
    st.close()
    assert worst[:12].max() < 2e-4 and worst[12:].max() < 1e-3, worst
    assert lifted >= 3, 'the scooper carries the forearm in the crafted states'


def test_coop_matches_oracle(am, device_fall):
    from assistive_gym_amd.libagx import Stepper
    from oracle_lib import Oracle
    coop = am.coop()
    o = Oracle(coop)
    states, _ = _states(coop, 8, 7301, arm_settler=device_fall)
    st = Stepper(coop, 8)
    st.set_state(states)
    worst = np.zeros(8)
    for k in range(4):
        act = np.random.RandomState(200 + k).uniform(-1, 1, (8, 24)).astype(np.float32)
        ref = st.get_state()
        _check_step(coop, o, st, ref, act, worst)
    st.close()
    assert worst.max() < 2e-4, worst


def test_vec_env_rollout_and_scalar_env(am):
    import torch
    from assistive_gym_amd.envs import make
    from assistive_gym_amd.vec_env import ArmManipulationSawyerVecEnv
    n = 64
    env = ArmManipulationSawyerVecEnv(n, pool_size=8, seed=3)
    obs = env.reset()
    assert obs.shape == (n, 45) and env.act_dim == 14
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, 14), device='cuda', generator=g) * 2 - 1)
        assert bool(done.all()) == (k == 199)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and env.stepper.overflow_count() == 0
    env.close()
    e = make('assistive_gym:ArmManipulationSawyerHuman-v1')
    o = e.reset()
    assert o['robot'].shape == (45,) and o['human'].shape == (42,)
    o, r, d, info = e.step({'robot': e.action_space_robot.sample(), 'human': e.action_space_human.sample()})
    assert np.isfinite(r['robot']) and not d['__all__'] and info['robot']['obs_human_len'] == 42 and info['robot']['action_robot_len'] == 14
    e.disconnect()
    e = make('ArmManipulationSawyer-v1')
    o = e.reset()
    assert o.shape == (45,)
    e.disconnect()


@pytest.mark.parametrize('robot', ['jaco', 'panda', 'baxter', 'pr2'])
def test_other_single_arm_robots(robot):
    """ArmManipulationJaco-v1 / ArmManipulationPanda-v1: pool states built the product way (both settles + collision rejection on the
    device), single steps against the oracle, a batched rollout"""
    import torch
    from assistive_gym_amd import libagx, vec_env
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from oracle_lib import Oracle
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    b = ModelBlob.load('arm_manipulation_' + robot)
    o = Oracle(b)
    n = 12
    states = vec_env.build_reset_pool(b, n, 9001)
    assert np.isfinite(states).all()
    st = Stepper(b, n)
    assert st.variant() == ('arm_manipulation_l' if robot in ('baxter', 'pr2') else 'arm_manipulation')      # two-armed robots: both arms dynamic, two tools
    st.set_state(states)
    worst = np.zeros(n)
    for k in range(3):
        act = np.random.RandomState(300 + k).uniform(-1, 1, (n, 14)).astype(np.float32)
        ref = st.get_state()
        _check_step(b, o, st, ref, act, worst)
    st.close()
    assert worst.max() < 1e-3, worst
    env = getattr(vec_env, 'ArmManipulation%sVecEnv' % {'pr2': 'PR2'}.get(robot, robot.capitalize()))(32, pool_size=8, seed=3)
    obs = env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((32, 14), device='cuda', generator=g) * 2 - 1)
    assert bool(done.all()) and torch.isfinite(obs).all() and torch.isfinite(rew).all() and env.stepper.overflow_count() == 0
    env.close()
