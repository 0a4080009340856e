#!/bin/bash
# round 5, session 25: the final tree (sweep lists as bytes) -- the GPU tests of every kernel variant's collision path, a 300-step bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R
export TMPDIR=/tmp
AGX_CONDITIONING_REPORT=$O/conditioning_tally_gpu.json timeout 320 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bed_bathing.py tests/test_gpu_scratch_itch.py tests/test_gpu_arm_manipulation.py tests/test_gpu_bench_size.py tests/test_gpu_solve_variants.py tests/test_golden_tasks.py tests/test_reference_pinned.py tests/test_gpu_feeding_robots.py -m gpu -q > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_subset.log; grep -E "^FAILED|passed|failed|oracle comparisons" $O/pytest_gpu_subset.log | tail -6 | cut -c1-300
timeout 60 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_300.json 2>$O/bench_300.err; python -c "
import json; j=json.loads(open('$O/bench_300.json').read().strip().splitlines()[-1]); print('300 steps', round(j['value']), j['ms_per_step'], 'solve ms per launch', j['roofline']['kernel_ms_per_launch'])"
