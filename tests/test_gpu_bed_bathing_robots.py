"""-m gpu: BedBathing<Robot>-v1 for Jaco, Panda, PR2 and Baxter on the HIP stepper (bed_bathing / bed_bathing_l kernel variants, through the
C ABI) against the CPU oracle, from pool states built the product way (rag-doll settle + collision rejection on the device).
PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['jaco', 'panda', 'pr2', 'baxter'])
def rb(request):
    from assistive_gym_amd import libagx
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    b = ModelBlob.load('bed_bathing_' + request.param)
    return request.param, b, Oracle(b)


def test_pool_states_and_single_steps_match_the_oracle(rb):
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    from test_scratch_itch_robots import flags_from_oracle
    name, b, o = rb
    n = 16
    states = build_reset_pool(b, n, 8001)
    flags = np.array([flags_from_oracle(b, o, s) for s in states])
    assert (flags & 1).sum() <= 2, flags                      # init_robot_pose's loop leaves at most the placements that failed three times
    st = Stepper(b, n)
    assert st.variant() == ('bed_bathing_l' if name == 'pr2' else 'bed_bathing')
    st.set_state(states)
    worst, touching = np.zeros(n), np.zeros(n, dtype=bool)
    for k in range(4):
        act = np.random.RandomState(100 + k).uniform(-1, 1, (n, 7)).astype(np.float32)
        ref = st.get_state()                                   # single-step comparison from the device's own state
        obs, rew, done, info = st.step_host(act)
        for i in range(n):
            o_obs, o_rew, o_done, o_info = o.step(ref[i], act[i])
            assert info[i, 6] == o_info[6] and abs(info[i, 7] - o_info[7]) <= 2, (i, info[i], o_info)
            dev = np.abs(obs[i] - o_obs)
            assert dev[-1] <= 1e-3 * max(1.0, abs(o_obs[-1]))
            dev[-1] = 0
            worst[i] = max(worst[i], float(dev.max()), abs(float(rew[i]) - o_rew) / max(1.0, abs(o_rew)))
            touching[i] |= o_info[6] > 0
            assert info[i, 4] == o_info[4] and info[i, 1] == o_info[1] and bool(done[i]) == o_done
            for c in (0, 2, 3):
                assert abs(info[i, c] - o_info[c]) <= 1e-3 * max(1.0, abs(o_info[c])), (i, c, info[i], o_info)
    st.close()
    # the bounds of tests/test_gpu_bed_bathing.py: 1e-4 in free space, 1e-3 for environments with contacts (forces agree to 1e-3 relative)
    assert worst[~touching].max(initial=0) < 1e-4 and worst[touching].max(initial=0) < 1e-3, (worst, touching)


def test_vec_env_rollout(rb):
    import torch
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.envs import make
    name, b, o = rb
    cls = getattr(vec_env, 'BedBathing%sVecEnv' % {'pr2': 'PR2'}.get(name, name.capitalize()))
    n = 64
    env = cls(n, pool_size=8, seed=3)
    obs = env.reset()
    assert obs.shape == (n, 24)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, 7), device='cuda', generator=g) * 2 - 1)
        assert bool(done.all()) == (k == 199)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and env.stepper.overflow_count() == 0
    env.close()
    if name == 'jaco':
        e = make('assistive_gym:BedBathingJacoHuman-v1')
        ob = e.reset()
        assert ob['robot'].shape == (24,) and ob['human'].shape == (28,)
        ob, r, d, info = e.step({'robot': e.action_space_robot.sample(), 'human': e.action_space_human.sample()})
        assert np.isfinite(r['robot']) and not d['__all__']
        e.disconnect()
