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


def test_wiping_pool_puts_the_pad_on_the_arm():
    """bench.py --task bedbathing --workload wiping (BASELINE config 3, 'dense tool-skin contact'): in every pool state the wiping pad
    touches the person's arm (a tool <-> human contact in the oracle's collision pass)"""
    import bench
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    b = ModelBlob.load('bed_bathing_sawyer')
    st = bench.wiping_pool(b, 6, 1001)
    o = Oracle(b)
    hits = 0
    for s in st:
        con = o.collide(s)
        tags = [(b.collider(int(c[0]))['tag'], b.collider(int(c[1]))['tag']) for c in con]
        hits += any(set(t) == {2, 3} for t in tags)            # AGX_TAG_TOOL, AGX_TAG_HUMAN
    assert hits == len(st), (hits, len(st))
