"""-m gpu: `python bench.py --gpus 1 --steps K --warmup W` exactly as the driver runs it at round end (small batch, so that it takes
seconds): exit code 0, ONE JSON line with the contract's keys, `roofline` and `cpu_baseline`, and every entry of "configs" -- each BASELINE
configuration that fits one GPU, including the scripted dense-contact workload, goes through the same run_config()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_command_small_batch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '2', '--envs-per-gpu', '256', '--pool', '16'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['metric'] == 'env_steps_per_sec' or 'env' in j['metric']
    assert j['n_gpus'] == 1 and j['steps'] == 6 and j['warmup'] == 2 and j['value'] > 0
    assert set(j['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(j['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    cfg = j['configs']
    for key in ('config3_BedBathingSawyer-v1', 'config3_BedBathingSawyer-v1_wiping_contact', 'config3_BedBathingSawyer-v1_dense',
                'config4_ScratchItchPR2Human-v1_1gpu', 'config5_DressingBaxter-v1_1gpu'):
        assert key in cfg and cfg[key]['value'] > 0, key
