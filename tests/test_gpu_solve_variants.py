"""-m gpu: the row-local Gauss-Seidel sweeps of the feeding variant compute the same bits (and so do the two vertex scans of the GJK support function:
lib/variants/scan4.so is the build with the 4-per-round scan of rounds 3-5, csrc/agx_gjk.h AGX_GJK_SCAN_WIDE).
  * csrc/agx_pgs_lvw.h (the default since round 6): up to four rows with disjoint velocity slots per visit, one per 16-lane group, list-scheduled
    per substep -- rows that share no slot commute exactly, rows that do keep their order;
  * csrc/agx_pgs_lvs.h (the default of round 5, now the fallback): one row per visit, row headers through scalar loads -- built by
    __graft_entry__.build() as lib/variants/lvs.so, and reachable inside the default build through the blob switch AGX_P_SOLVE_WIDE = 0;
  * csrc/agx_pgs_lv.h (-DAGX_PGS_LV=2: headers, impulses and velocity slots in LDS; lib/variants/lv2.so).
40 steps of 1,024 FeedingJaco environments (pool resets included) end in BIT-IDENTICAL states, observations, rewards and info words, whatever the
size of the LDS window (rows beyond it read their pairs from the scratch record).  Each build runs in its own process (AGX_LIB)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'gpu_lv_bits.py')
LV2 = os.path.join(ROOT, 'assistive_gym_amd', 'lib', 'variants', 'lv2.so')
LVS = os.path.join(ROOT, 'assistive_gym_amd', 'lib', 'variants', 'lvs.so')
SCAN4 = os.path.join(ROOT, 'assistive_gym_amd', 'lib', 'variants', 'scan4.so')


def _rollout(out, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, TOOL, out, '1024', '40'], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]


def test_row_local_sweeps_bit_identical(tmp_path):
    assert os.path.exists(LV2) and os.path.exists(LVS) and os.path.exists(SCAN4), 'lib/variants/{lv2,lvs,scan4}.so are missing: run __graft_entry__.build()'
    runs = {'wide (default)': {}, 'narrow build (lvs.so)': {'AGX_LIB': LVS}, 'headers in LDS (lv2.so)': {'AGX_LIB': LV2},
            'wide, smallest window': {'AGX_SOLVE_LDS_BYTES': '9536'},          # (the smallest solve launch: most friction rows lie beyond the window)
            'wide build, SOLVE_WIDE = 0': {'AGX_BITS_PARAM': 'SOLVE_WIDE=0'}, 'wide, 12 KB': {'AGX_SOLVE_LDS_BYTES': '12288'},
            'support scan of rounds 3-5 (scan4.so)': {'AGX_LIB': SCAN4}}
    paths = {}
    for k, (name, env) in enumerate(runs.items()):
        paths[name] = str(tmp_path / ('run%d.npz' % k))
        _rollout(paths[name], env)
    ref = paths['wide (default)']
    for name, p in paths.items():
        if p != ref:
            r = subprocess.run([sys.executable, TOOL, '--compare', ref, p], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and 'IDENTICAL' in r.stdout, (name, r.stdout[-1500:])
