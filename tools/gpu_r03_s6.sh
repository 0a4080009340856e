#!/bin/bash
# round 3, GPU session 6: cloth kernel with patch-local link relaxation (no workgroup barrier for the links inside a wave's patch)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03j}
rm -rf $O && mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_dressing.py tests/test_reference_pinned.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -25) > $O/gputest_dressing.log; tail -3 $O/gputest_dressing.log
timeout 300 python bench.py --task dressing --steps 30 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_dressing.json 2> $O/bench_dressing.err
python - <<PY
import json
try:
    j = json.load(open('$O/bench_dressing.json')); print('dressing', round(j['value']), j['ms_per_step'])
except Exception as e: print('dressing failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dressing -- python $R/bench.py --task dressing --steps 10 --warmup 2 --no-cpu-baseline --no-configs > $O/bench_dressing_under_rocprof.json 2> $O/stats_dressing.err
f=$(find $O/stats_dressing -name "*kernel_stats.csv" | head -1); head -6 $f | cut -d, -f1-6
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dressing_unchunked -- python $R/bench.py --task dressing --steps 6 --warmup 2 --no-cpu-baseline --no-configs > $O/bench_dressing_unchunked_under_rocprof.json 2> $O/stats_dressing_unchunked.err
f=$(find $O/stats_dressing_unchunked -name "*kernel_stats.csv" | head -1); head -6 $f | cut -d, -f1-6
