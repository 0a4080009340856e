#!/bin/bash
# round 5, session 19: chunk count and LDS size of the solve launch again, with the 32-byte row headers
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R
export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for L in 9536 10240 11264 12288; do AGX_SOLVE_LDS_BYTES=$L timeout 300 $B > $O/bench_lds$L.json 2>/dev/null; line lds$L < $O/bench_lds$L.json | tee -a $O/ab.txt; done
for C in 2 3 4 5 6; do AGX_CHUNKS=$C timeout 300 $B > $O/bench_c$C.json 2>/dev/null; line chunks$C < $O/bench_c$C.json | tee -a $O/ab.txt; done
for C in 4 6; do GPU_MAX_HW_QUEUES=8 AGX_CHUNKS=$C timeout 300 $B > $O/bench_c${C}_q8.json 2>/dev/null; line chunks${C}_hwq8 < $O/bench_c${C}_q8.json | tee -a $O/ab.txt; done
