#!/bin/bash
# round 4, session 19: chunk-stream sweep of the final kernels at 8 hardware queues (bench.py's setting): 2, 3, 4, 6 chunks, 300 steps each, twice
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
for rep in 1 2; do for c in 2 3 4 6; do
  AGX_CHUNKS=$c timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs > $O/chunks${c}_$rep.json 2> $O/chunks${c}_$rep.err
  python - <<PY
import json
try:
    j = json.load(open('$O/chunks${c}_$rep.json')); print('chunks $c', round(j['value']), {k[4:-7]: round(x, 2) for k, x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})
except Exception as e: print('chunks $c failed', e)
PY
done; done 2>&1 | tee $O/chunks_sweep.txt
