#!/bin/bash
# on the GPU box: bench every variant twice, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in assistive_gym_amd/lib/variants/*.so; do
AGX_LIB=$PWD/$v timeout 300 python bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v'.split('/')[-1], round(j['value']), {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"
done; done
