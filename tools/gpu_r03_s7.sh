#!/bin/bash
# round 3, GPU session 7: row-space solve with a triangular coupling matrix (8 instead of 5 environments per CU): configs 3 / 4 / arm manipulation
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03l}
rm -rf $O && mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_bed_bathing.py tests/test_gpu_scratch_itch.py tests/test_gpu_arm_manipulation.py tests/test_golden_tasks.py tests/test_reference_pinned.py -m gpu -q -x 2>&1 | tail -15) > $O/gputest.log; tail -3 $O/gputest.log
for t in bedbathing scratchitch armmanipulation; do
  timeout 300 python bench.py --task $t --steps 600 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_$t.json 2> $O/bench_$t.err
done
timeout 300 python bench.py --task bedbathing --workload wiping --steps 600 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_wiping.json 2> $O/bench_wiping.err
python - <<PY
import json
for t in ('bedbathing', 'wiping', 'scratchitch', 'armmanipulation'):
    try:
        j = json.load(open('$O/bench_%s.json' % t)); print(t, round(j['value']), j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'], j['contacts_per_substep'])
    except Exception as e: print(t, 'failed', e)
PY
cd /tmp && export TMPDIR=/tmp
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_bedbathing -- python $R/bench.py --task bedbathing --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof_bedbathing.json 2> $O/stats_bed.err
f=$(find $O/stats_unchunked_bedbathing -name "*kernel_stats.csv" | head -1); head -5 $f | cut -d, -f1-6
