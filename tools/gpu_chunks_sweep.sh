#!/bin/bash
# env-steps/s of the headline workload against the number of chunk streams (AGX_CHUNKS)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03m}
rm -rf $O && mkdir -p $O
cd $R
for c in 1 2 3 4 5 6 8; do
  AGX_CHUNKS=$c timeout 200 python bench.py --task ${2:-feeding} --steps 300 --warmup 20 --no-cpu-baseline --no-configs > $O/chunks$c.json 2> $O/chunks$c.err
  python - <<PY
import json
try:
    j = json.load(open('$O/chunks$c.json')); print('chunks $c', round(j['value']), j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'])
except Exception as e: print('chunks $c failed', e)
PY
done
