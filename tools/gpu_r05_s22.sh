#!/bin/bash
# round 5, session 22: the final tree (rows beyond the window wait for their own pair only) -- full GPU suite, the bench lines (default with all configs,
# the driver's command), kernel traces
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
export TMPDIR=/tmp
AGX_CONDITIONING_REPORT=$O/conditioning_tally_gpu.json timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|passed|failed|oracle comparisons" $O/pytest_gpu.log | tail -14
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; python -c "
import json; j=json.loads(open('$O/driver_cmd.json').read().strip().splitlines()[-1]); print('driver cmd', round(j['value']), j['ms_per_step'], 'roofline frac', j['roofline']['frac'], 'cpu', j.get('cpu_baseline',{}).get('value'))"
timeout 900 python bench.py > $O/bench_default_all_configs.json 2>$O/bench_default.err; python -c "
import json; j=json.loads(open('$O/bench_default_all_configs.json').read().strip().splitlines()[-1]); print('default 2000 steps', round(j['value']), j['ms_per_step']); [print('  ', k, round(v['value']), v.get('contacts_per_substep')) for k,v in j.get('configs',{}).items()]"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_under_rocprof_feeding.json 2>$O/rocprof.err ); find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_feeding.csv; head -6 $O/kernel_stats_feeding.csv; rm -rf $O/prof
( cd /tmp && AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof_feeding.json 2>>$O/rocprof.err ); find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_unchunked_feeding.csv; head -5 $O/kernel_stats_unchunked_feeding.csv; rm -rf $O/prof1
timeout 300 python tools/gpu_rllib_overhead.py 2>/dev/null | tail -1 | tee $O/rllib_overhead.json
