"""Consumer of the parity protocol's reference dumps.  `tools/pybullet_dump.py`, run
where the reference's PyBullet fork exists, records per step the full state (in this repo's state-record
layout), the action and the reference's observation / reward / done / total_force_on_human.  Here every
recorded state is injected, the recorded action applied, and the outputs compared (1e-3 relative, the
bound BASELINE.json's north_star names; absolute floor 1e-4).

No such dump can be produced in the build container (no pybullet): the tests SKIP until a file
tests/golden/pybullet_dump_*.npz is committed -- until then parity is unpinned."""
import glob
import os

import numpy as np
import pytest

DUMPS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pybullet_dump_*.npz')))
REL, FLOOR = 1e-3, 1e-4


def _check(name, got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), FLOOR / REL)
    assert err.max() < REL, '%s deviates from the reference dump: max rel. error %.3g' % (name, err.max())


def _load(path, blob):
    d = np.load(path, allow_pickle=False)
    if 'model' in d.files and str(d['model']) != 'feeding_jaco':          # a dump of another Feeding<Robot>-v1 (tools/pybullet_dump.py --env)
        from assistive_gym_amd.blob import ModelBlob
        blob = ModelBlob.load(str(d['model']))
    assert int(d['blob_version']) == blob.h['VERSION'], 'dump was recorded for another blob version'
    assert d['states'].shape[1] == blob.state_words and len(d['states']) == len(d['actions']) + 1
    return d, blob


@pytest.mark.skipif(not DUMPS, reason='no PyBullet reference dump committed (tools/pybullet_dump.py needs the reference stack)')
@pytest.mark.parametrize('path', DUMPS)
def test_oracle_matches_reference_dump(path, blob, oracle):
    d, blob = _load(path, blob)
    if blob.words is not oracle.blob.words:
        from oracle_lib import Oracle
        oracle = Oracle(blob)
    for k in range(len(d['actions'])):
        s = d['states'][k].copy()
        obs, rew, done, info = oracle.step(s, d['actions'][k])
        _check('oracle obs @%d' % k, obs, d['obs'][k]); _check('oracle reward @%d' % k, rew, d['reward'][k])
        _check('oracle force @%d' % k, info[0], d['total_force_on_human'][k])
        assert bool(done) == bool(d['done'][k])


@pytest.mark.gpu
@pytest.mark.skipif(not DUMPS, reason='no PyBullet reference dump committed (tools/pybullet_dump.py needs the reference stack)')
@pytest.mark.parametrize('path', DUMPS)
def test_stepper_matches_reference_dump(path, blob):
    from assistive_gym_amd.libagx import Stepper
    d, blob = _load(path, blob)
    T = len(d['actions'])
    st = Stepper(blob, T)                     # step k of the episode runs in environment slot k
    st.set_state(d['states'][:T])
    obs, rew, done, info = st.step_host(d['actions'])
    for k in range(T):
        _check('obs @%d' % k, obs[k], d['obs'][k]); _check('reward @%d' % k, rew[k], d['reward'][k])
        _check('force @%d' % k, info[k, 0], d['total_force_on_human'][k])
        assert bool(done[k]) == bool(d['done'][k])
    st.close()
