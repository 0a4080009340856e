"""-m gpu: the two row-local Gauss-Seidel sweeps of the feeding variant -- csrc/agx_pgs_lvs.h (the default: row headers through scalar loads,
impulses in a vector register, 10 KB of LDS) and csrc/agx_pgs_lv.h (-DAGX_PGS_LV=2: headers, impulses and velocity slots in LDS, 20 KB; built
by __graft_entry__.build() as lib/variants/lv2.so) -- visit the same rows in the same order with the same arithmetic: 40 steps of 1,024
FeedingJaco environments (episode boundaries included) end in BIT-IDENTICAL states, observations, rewards and info words, whatever the
size of the LDS window (rows beyond it read their pairs from the scratch record).  Each build runs in its own process (AGX_LIB)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'gpu_lv_bits.py')
LV2 = os.path.join(ROOT, 'assistive_gym_amd', 'lib', 'variants', 'lv2.so')


def _rollout(out, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, TOOL, out, '1024', '40'], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]


def test_row_local_sweeps_bit_identical(tmp_path):
    assert os.path.exists(LV2), 'lib/variants/lv2.so is missing: run __graft_entry__.build()'
    a, b, c = (str(tmp_path / n) for n in ('default.npz', 'lv2.npz', 'default_small_window.npz'))
    _rollout(a, {})
    _rollout(b, {'AGX_LIB': LV2})
    _rollout(c, {'AGX_SOLVE_LDS_BYTES': '9536'})          # (the smallest solve launch: a fifth of the pairs lie beyond the window)
    for other in (b, c):
        r = subprocess.run([sys.executable, TOOL, '--compare', a, other], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and 'IDENTICAL' in r.stdout, r.stdout[-1500:]
