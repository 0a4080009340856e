"""-m gpu: FeedingStretch-v1, ScratchItchStretch-v1 and BedBathingStretch-v1 on the HIP stepper (the *_m kernel variants: 16 robot DoFs on a
floating base) against the CPU oracle, from pool states built the product way (numpy mobile-base sampler + the device's collision pass and
settles).  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


CASES = [('feeding', False), ('feeding', True), ('scratch_itch', False), ('scratch_itch', True), ('bed_bathing', False), ('bed_bathing', True)]
VEC = {'feeding': 'FeedingStretchVecEnv', 'scratch_itch': 'ScratchItchStretchVecEnv', 'bed_bathing': 'BedBathingStretchVecEnv'}
IDS = {'feeding': 'FeedingStretch', 'scratch_itch': 'ScratchItchStretch', 'bed_bathing': 'BedBathingStretch'}


@pytest.fixture(scope='module', params=CASES, ids=['%s-%s' % (t, 'coop' if c else 'robot') for t, c in CASES])
def rb(request):
    from assistive_gym_amd import libagx
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    task, coop = request.param
    b = ModelBlob.load(task + '_stretch')
    if coop:
        b = b.coop()
    b.task_name = task
    return b, Oracle(b)


def test_step_matches_oracle(rb):
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    b, oracle = rb
    n, steps = 16, 4
    # bed bathing: a trembling arm lying on the bed is a chaotic contact state (a 1e-6 change of one joint angle moves the oracle's own
    # result by 4e-4 within one step); tests/test_gpu_bed_bathing.py covers the tremor path, here the robot is what is new
    states = build_reset_pool(b, n, 5001, impairment='no_tremor' if b.task_name == 'bed_bathing' else 'random')
    assert np.isfinite(states[:, :b.h['S_ENV']]).all()
    st = Stepper(b, n)
    assert st.variant() == b.task_name + '_m'
    rng = np.random.RandomState(7)
    ref = states.copy()
    worst = dict(obs=0.0, reward=0.0, force=0.0, q=0.0)
    flips = 0
    for k in range(steps):
        st.set_state(ref)                   # single-step parity
        actions = rng.uniform(-1, 1, (n, b.act_dim)).astype(np.float32)
        obs, rew, done, info = st.step_host(actions)
        got = st.get_state()
        for i in range(n):
            o_obs, o_rew, o_done, o_info = oracle.step(ref[i], actions[i])
            worst['obs'] = max(worst['obs'], np.abs(obs[i] - o_obs).max())
            worst['reward'] = max(worst['reward'], abs(rew[i] - o_rew) / max(1.0, abs(o_rew)))
            worst['force'] = max(worst['force'], abs(info[i, 0] - o_info[0]) / max(1.0, abs(o_info[0])))
            worst['q'] = max(worst['q'], np.abs(b.view(got[i])['q'] - b.view(ref[i])['q']).max())
            assert bool(done[i]) == o_done
            flips += int(info[i, 6] != o_info[6])
    st.close()
    print('worst deviations', worst, 'contact-count flips', flips, 'of', n * steps)
    assert flips <= 0.08 * n * steps
    assert worst['obs'] < 1e-4 and worst['reward'] < 1e-4 and worst['force'] < 1e-3 and worst['q'] < 5e-5


def test_driving_on_the_device_follows_the_oracle(rb):
    """two seconds of driving (both wheels forward, then a turn) from one settled state: the base pose of the device run stays within a
    millimetre / a milliradian of the oracle's, and the robot stays on the ground"""
    from assistive_gym_amd.libagx import Stepper
    from assistive_gym_amd.vec_env import build_reset_pool
    b, oracle = rb
    n = 4
    states = build_reset_pool(b, n, 5003, impairment='none')
    for i in range(n):
        b.view(states[i:i + 1])['plane_friction'][0] = 0.5
    st = Stepper(b, n)
    st.set_state(states)
    ref = states.copy()
    a = np.zeros((n, b.act_dim), np.float32)
    for k in range(20):
        a[:, 0], a[:, 1] = 1.0, (1.0 if k < 10 else -0.5)
        st.step_host(a)
        for i in range(n):
            oracle.step(ref[i], a[i])
    got = st.get_state()
    st.close()
    for i in range(n):
        qd, qo = b.view(got[i:i + 1])['q'][0], b.view(ref[i:i + 1])['q'][0]
        assert np.linalg.norm(qo[:2]) > 0.1 and abs(qo[3]) > 0.1                  # it drove and turned
        assert np.abs(qd[:6] - qo[:6]).max() < 3e-2, (qd[:6], qo[:6])      # 100 substeps of slipping wheel contacts, f32 against f64: a few per cent of the 0.2 m / 0.7 rad driven (single-step parity above)
        assert abs(qd[2] + 0.09) < 3e-3 and np.all(np.abs(qd[4:6]) < 1e-2)


def test_vec_env_rollout_and_scalar_env(rb):
    import torch
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.envs import make
    b, oracle = rb
    n = 64
    env = getattr(vec_env, VEC[b.task_name])(n, pool_size=8, seed=3, coop=b.is_coop)
    obs = env.reset()
    assert obs.shape == (n, b.obs_dim)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in range(200):
        obs, rew, done, info = env.step(torch.rand((n, b.act_dim), device='cuda', generator=g) * 2 - 1)
        assert bool(done.all()) == (k == 199)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert int((info[:, 6] >= 1.0e6).sum()) == 0                                   # no environment tripped the non-finite guard (AGX_INFO_NONFINITE)
    assert env.stepper.overflow_count() < 0.06 * n * 200 * 5                       # contacts dropped by the 64-contact budget: the spoon often rests on the table here
    env.close()
    e = make('assistive_gym:%s%s-v1' % (IDS[b.task_name], 'Human' if b.is_coop else ''))
    o = e.reset()
    if b.is_coop:
        assert o['robot'].shape[0] + o['human'].shape[0] == b.obs_dim
    else:
        assert o.shape == (b.obs_dim,)
    e.disconnect()
