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
    import conditioning as C
    for i in range(len(ref)):
        start = ref[i].copy()
        o_obs, o_rew, o_done, o_info = o.step(ref[i], act[i])
        assert info[i, 6] == o_info[6] and abs(info[i, 7] - o_info[7]) <= 2, (i, info[i], o_info)
        dev = np.abs(obs[i] - o_obs)
        forces = [r - 2, r - 1] + ([blob.obs_dim - 3, blob.obs_dim - 2, blob.obs_dim - 1] if blob.is_coop else [])
        cache = {}
        def sens(eps, _i=i, _start=start):                 # the oracle's own response to input perturbations, evaluated only for a case beyond 1e-3
            if eps not in cache:
                cache[eps] = C.ulp_sensitivity(blob, o, _start, act[_i], trials=4 if eps is None else 6, rel_eps=eps)
            return cache[eps]
        for k in forces:
            # north_star's 1e-3; beyond it the measured bounds of tests/conditioning.py (a sampled start with the scooper pressed into the
            # mattress: 21 N in session r04i)
            ok, lim, sv = C.within(dev[k], o_obs[k], lambda: sens(None)['obs'][k], floor=C.force_floor(blob), step_sens_fn=lambda: sens(C.STEP_EPS)['obs'][k],
                                   geom_sens_fn=lambda: sens(C.GEOM_EPS)['obs'][k])
            if sv is not None:
                print('conditioned: env %d force entry %d: device %.5g oracle %.5g, bound %.3g' % (i, k, obs[i, k], o_obs[k], lim))
            if not ok:
                # a VIOLENT step (tests/test_reference_pinned.py): a sampled start that PENETRATES (the scooper more than a millimetre inside the
                # mattress: the re-draws of the reset reject contacts with the person and the furniture COLLISION_TRIES times, then accept) is
                # pushed out within one substep -- no penetration-recovery clamp, DESIGN 2 -- with tens of newtons; judged against the oracle
                # under a 1e-5 relative perturbation of its input (session r04j: ArmManipulationJaco, 21.01 N vs 20.88 N)
                con = o.collide(start.copy())
                assert len(con) and con[:, 11].min() < -1e-3, (i, k, obs[i, k], o_obs[k], lim, sv)
                lim = max(lim, C.K * sens(1e-5)['obs'][k])
                print('VIOLENT STEP: env %d starts %.2f mm inside a collider: force entry %d device %.5g oracle %.5g, bound %.3g' % (i, -1e3 * con[:, 11].min(), k, obs[i, k], o_obs[k], lim))
                ok = dev[k] <= lim
            assert ok, (i, k, obs[i, k], o_obs[k], lim, sv)
            dev[k] = 0
        # the reward carries 0.01 * pressure and 0.01 * forces: compare it at the forces' tolerance
        worst[i] = max(worst[i], float(dev.max()), abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)) * 0.1)
        assert info[i, 4] == o_info[4] and info[i, 1] == o_info[1] and bool(done[i]) == o_done
        for c in (0, 2, 3):
            ok, lim, sv = C.within(abs(info[i, c] - o_info[c]), o_info[c], lambda: sens(None)['info'][c], floor=C.force_floor(blob), step_sens_fn=lambda: sens(C.STEP_EPS)['info'][c],
                                   geom_sens_fn=lambda: sens(C.GEOM_EPS)['info'][c])
            assert ok, (i, c, info[i], o_info, lim, sv)
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


@pytest.mark.gpu
@pytest.mark.parametrize('robot', ['sawyer', 'jaco', 'panda', 'pr2', 'baxter'])
def test_reset_on_the_device_three_models_in_a_row(robot):
    """ArmManipulationEnv.reset through the C ABI (arm_manipulation.py:110-180): agx_sample_reset on the task's handle runs the rag doll's drop
    and 100-step settle (bed_settle), the fall model's sampler and the arm's 100-step fall, then the task's sampler with collision rejection.
    The two intermediate records are read back and handed to the numpy restatement (oracle/reset_oracle.py): it must arrive at the same
    post-reset record; the records must be steppable."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import attach_arm_fall_models
    from test_reset_generator import assert_same_record
    b = ModelBlob.load('arm_manipulation_' + robot)
    n, seed = 24, 91000
    st = Stepper(b, n)
    fall, rag = attach_arm_fall_models(st, b, n, 0)
    st.sample_reset(seed, impairment='no_tremor')
    st.synchronize()
    got, fell, lying = st.get_state(), fall.get_state(), rag.get_state()
    assert np.isfinite(got[:, :b.h['S_ENV']]).all() and np.isfinite(fell[:, :b.h['S_ENV']]).all()
    v, fv = b.view(got), b.view(fell)
    nr = b.nrobot
    # the arm fell (it started at shoulder 60 / -60 degrees, arm_manipulation.py:139) and the robot left its parking spot
    dyn = b.meta['human_dynamic_joints']
    assert np.abs(fv['q'][:, nr + dyn.index(3)] - np.deg2rad(60)).max() > 0.05
    assert np.abs(fv['base'][:, 0] - 20.0).max() < 1e-5 and np.abs(v['base'][:, :2]).max() < 3.0
    assert np.array_equal(v['human'], fv['human']) and np.array_equal(v['q'][:, nr:], fv['q'][:, nr:]) and np.array_equal(v['qd'][:, nr:], fv['qd'][:, nr:])
    R = ro.with_collision_check(b.words)
    ok = 0
    for i in range(6):
        want, info = R.sample(seed + i, impairment_mode=ro.MODE_NO_TREMOR, settled=lying[i], fell=fell[i])
        assert_same_record(b, want, got[i], '%s env %d' % (robot, i))
        ok += int(info['ik_ok'])
    assert ok >= 4
    flags = st.check_collisions()
    # AGX_COLLIDE_ENV: the arm or the scooper touching the person / the bed -- what the re-draws are for (a few placements keep colliding
    # after COLLISION_TRIES of them, as on the host path; self-interpenetration of the arm, bit 1, is not a reason to re-draw: env.py:299-308)
    assert ((flags & 1) == 0).mean() > 0.6, flags
    rng = np.random.RandomState(1)
    for k in range(10):
        obs, rew, done, info = st.step_host(rng.uniform(-1, 1, (n, b.act_dim)).astype(np.float32))
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    st.close(); fall.close(); rag.close()


@pytest.mark.gpu
def test_vec_env_with_device_resets():
    """ArmManipulationSawyerVecEnv(reset='device'): every episode of every environment starts from a newly dropped, fallen and placed scene"""
    import torch
    from assistive_gym_amd.vec_env import ArmManipulationSawyerVecEnv
    env = ArmManipulationSawyerVecEnv(48, reset='device', seed=77)
    obs = env.reset()
    first = env.stepper.get_state().copy()
    assert obs.shape == (48, env.blob.obs_dim) and bool(torch.isfinite(obs).all())
    for k in range(4):
        obs, rew, done, info = env.step(torch.zeros(48, env.blob.act_dim, device=obs.device))
    assert bool(torch.isfinite(obs).all())
    s = env._stream()
    env._fresh_reset(None, s)                             # what step() does at the 200-step boundary (there with the done mask)
    torch.cuda.synchronize()
    second = env.stepper.get_state()
    v1, v2 = env.blob.view(first), env.blob.view(second)
    assert np.abs(v1['human'] - v2['human']).max() > 1e-3 and np.abs(v1['base'] - v2['base']).max() > 1e-3          # another human lies there, the robot stands elsewhere
    assert (v2['iteration'] == 0).all() and np.isfinite(second[:, :env.blob.h['S_ENV']]).all()
    obs, rew, done, info = env.step(torch.zeros(48, env.blob.act_dim, device=obs.device))
    assert bool(torch.isfinite(obs).all())
    env.close()
