"""DrinkingJaco on the device -- the `drinking` kernel variant through the C ABI against what the reference's own DrinkingJacoEnv.step()
returned (tests/golden/ref_steps.npz).  Written in round 3 WITHOUT a GPU at hand (the variant was checked on the CPU wave emulator:
test_reference_pinned.py::test_emulator_drinking_step_matches_the_reference); the file sorts last so that a failure here cannot hide the
rest of a `-x` run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cases():
    import torch
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    import test_reference_pinned as T
    return T, [T.case(n) for n in T.NAMES if n.startswith('drinking')]


@pytest.mark.parametrize('coop', [False, True])
def test_gpu_drinking_step_matches_the_reference(cases, coop):
    from assistive_gym_amd import libagx
    from refcases import variant_blob
    T, cs = cases
    cs = [c for c in cs if c['coop'] == coop]
    b = variant_blob('drinking_jaco', coop, '')
    st = libagx.Stepper(b, len(cs))
    st.set_state(np.stack([c['state'] for c in cs]))
    st.set_cloth(np.stack([c['cloth'] for c in cs]))
    obs, rew, done, info = st.step_host(np.stack([c['action'] for c in cs]))
    out, water = st.get_state(), st.get_cloth()
    for i, c in enumerate(cs):
        T.check_step(b, c, obs[i], rew[i], done[i], info[i], tol=1e-4, ftol=1e-3)
        T.check_state(b, c, out[i], tol=1e-4)
        T.check_water(c['name'], water[i], tol=5e-4)
    st.close()


def test_gpu_drinking_episode_is_finite_and_pours(cases):
    """4 environments x 90 steps of the tipping action: every output finite, no environment keeps all its water, the masks only lose bits"""
    from assistive_gym_amd import libagx
    from refcases import variant_blob
    T, cs = cases
    c = [c for c in cs if c['name'] == 'drinking_none_step0'][0]
    b = variant_blob('drinking_jaco', False, '')
    n = 4
    st = libagx.Stepper(b, n)
    st.set_state(np.stack([c['state']] * n)); st.set_cloth(np.stack([c['cloth']] * n))
    a = np.zeros((n, 7), np.float32); a[:, 4], a[:, 5], a[:, 6] = 0.5, 1.0, 1.0
    t = b.h['S_TASK']
    prev = st.get_state().view(np.uint32)[:, t:t + 4].copy()
    total = np.zeros(n)
    for k in range(90):
        obs, rew, done, info = st.step_host(a)
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
        cur = st.get_state().view(np.uint32)[:, t:t + 4]
        assert not (cur & ~prev).any()
        prev = cur.copy(); total += info[:, 4]
    alive = np.array([sum(bin(int(x)).count('1') for x in row[:2]) for row in prev])
    assert (alive < 20).all() and np.allclose(total, -(64 - alive))
    st.close()
