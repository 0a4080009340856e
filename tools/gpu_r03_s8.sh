#!/bin/bash
# round 3, GPU session 8: resets -- config 4 with reset='device' (base pose search on the GPU), pool refresh rates of configs 3 and 5
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03n}
rm -rf $O && mkdir -p $O
cd $R
timeout 300 python bench.py --task scratchitch --steps 600 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_scratchitch_pool.json 2> $O/e1.err
timeout 300 python bench.py --task scratchitch --reset device --steps 600 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_scratchitch_device_reset.json 2> $O/e2.err
timeout 400 python bench.py --task bedbathing --pool-refresh 32 --steps 1000 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_bedbathing_pool_refresh.json 2> $O/e3.err
timeout 400 python bench.py --task dressing --pool-refresh 8 --steps 410 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_dressing_pool_refresh.json 2> $O/e4.err
python - <<PY
import json
for f in ('bench_scratchitch_pool', 'bench_scratchitch_device_reset', 'bench_bedbathing_pool_refresh', 'bench_dressing_pool_refresh'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value']), 'ms/step %.3f' % j['ms_per_step'], 'refreshed', j.get('pool_states_refreshed'), j['config']['reset'])
    except Exception as e: print(f, 'failed', e)
PY
python tools/gpu_toc_timing.py scratch_itch_pr2 4096 > $O/toc_timing_scratch_itch_pr2.json 2> $O/e5.err; cat $O/toc_timing_scratch_itch_pr2.json
