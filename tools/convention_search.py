"""Which of Bullet's solver conventions does a PyBullet dump follow?  (the one thing that can be prepared here.)

`tools/pybullet_dump.py`, run where the reference's PyBullet fork exists, records an episode -- per step the full state in this repository's
state-record layout, the action, and what the reference returned (observation, reward, done, total_force_on_human).  This script replays the
dump on the CPU oracle under every combination of the [BULLET-UNVERIFIED] switches of include/agx_blob.h -- warm start 0.85, second friction
direction, persistent 4-point manifold, split-impulse threshold 4 cm, residual early-out 1e-7 -- and ranks the combinations by their largest
relative deviation from the dump.  Every recorded state is injected before its step (so errors do not accumulate), while the solver's own
memory (warm-start impulses, cached manifold points) lives on from step to step, as Bullet's does.  The four device-side switches of the best
combination are then blob parameters for `agx_create`.

    python tools/convention_search.py tests/golden/pybullet_dump_<name>.npz [--top 8]

Test infrastructure (it drives the oracle); the product never imports it."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

SWITCHES = (('warm start 0.85', 'WARMSTART', 0.85), ('two friction directions', 'FRICTION_DIRS', 2.0), ('persistent manifold', 'MANIFOLD', 1.0),
            ('split impulse below 4 cm', 'SPLIT_PEN', 0.04), ('residual early-out 1e-7', 'ORACLE_RESIDUAL_EPS', 1e-7))
REL, FLOOR = 1e-3, 1e-4                      # north_star's bound and the absolute floor of tests/test_reference_dump.py


def rel_err(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float((np.abs(got - ref) / np.maximum(np.abs(ref), FLOOR / REL)).max())


def evaluate(blob, d, combo):
    """-> dict(obs=, reward=, force=, done_mismatches=): largest relative deviations of the oracle under `combo` (a tuple of booleans over
    SWITCHES) from the dump `d`"""
    from oracle_lib import Oracle
    b = blob
    for on, (_, key, val) in zip(combo, SWITCHES):
        if on:
            b = b.set_param(key, val)
    o = Oracle(b)
    o.forget_warm()
    out = dict(obs=0.0, reward=0.0, force=0.0, done_mismatches=0)
    has_cloth = 'cloth' in d.files
    for k in range(len(d['actions'])):
        s = d['states'][k].copy()
        if has_cloth:
            obs, rew, done, info = o.step_cloth(s, d['cloth'][k].copy(), d['actions'][k])
        else:
            obs, rew, done, info = o.step(s, d['actions'][k])
        out['obs'] = max(out['obs'], rel_err(obs, d['obs'][k])); out['reward'] = max(out['reward'], rel_err(rew, d['reward'][k]))
        out['force'] = max(out['force'], rel_err(info[0], d['total_force_on_human'][k]))
        out['done_mismatches'] += int(bool(done) != bool(d['done'][k]))
    o.forget_warm()
    return out


def search(path):
    """-> list of (combo, result) sorted by the largest of the three deviations"""
    from assistive_gym_amd.blob import ModelBlob
    d = np.load(path, allow_pickle=False)
    blob = ModelBlob.load(str(d['model']) if 'model' in d.files else 'feeding_jaco')
    assert int(d['blob_version']) == blob.h['VERSION'], 'the dump was recorded for another blob version: re-run tools/pybullet_dump.py'
    res = [(c, evaluate(blob, d, c)) for c in itertools.product((False, True), repeat=len(SWITCHES))]
    res.sort(key=lambda r: (max(r[1]['obs'], r[1]['reward'], r[1]['force']), sum(r[0])))
    return res


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 8
    res = search(path)
    print('| conventions switched on | obs | reward | total_force_on_human | done mismatches | within 1e-3 |')
    print('|---|---|---|---|---|---|')
    for c, r in res[:top]:
        names = ', '.join(n for on, (n, _, _) in zip(c, SWITCHES) if on) or '(the defaults)'
        worst = max(r['obs'], r['reward'], r['force'])
        print('| %s | %.2e | %.2e | %.2e | %d | %s |' % (names, r['obs'], r['reward'], r['force'], r['done_mismatches'], 'yes' if worst < REL and not r['done_mismatches'] else 'no'))


if __name__ == '__main__':
    main()
