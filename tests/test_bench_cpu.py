"""bench.py's host-side pieces that do not need a GPU: the usable-core detection and the CPU-oracle
worker processes of the `cpu_baseline` leg (kind = "port": the C restatement, never the product path)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_usable_cores():
    import bench
    n = bench._usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_cpu_baseline_leg(blob):
    import bench
    from assistive_gym_amd.host.reset import make_states
    st, _ = make_states(blob, 4, seed=11)
    r = bench.cpu_baseline(blob, st, 1, 3)
    assert r['kind'] == 'port' and r['unit'] == 'env-steps/s' and r['cores'] == bench._usable_cores()
    assert r['value'] > 0 and 'not PyBullet' in r['sample']
