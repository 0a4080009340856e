#!/bin/bash
# round 3, GPU session 5: cloth kernel with its impulse sums in HBM (LDS 156 -> 113 KB: rigid kernels of another chunk can share the CU),
# the whole -m gpu suite, dressing bench + kernel trace, the default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03i}
rm -rf $O && mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/gputest.log; tail -3 $O/gputest.log
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
cd $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
try:
    j = json.load(open('$O/bench_default.json')); print('default', round(j['value']), {k: round(v['value']) for k, v in j.get('configs', {}).items()})
except Exception as e: print('default failed', e)
PY
