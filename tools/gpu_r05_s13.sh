#!/bin/bash
# round 5, session 13: marginal cost of one more instruction of a kind inside a visit of the scalar-header row-local sweep (redundant instructions,
# same results): builds with 4 more VALU / SALU / s_nop, 2 more LDS reads / writes per visit -- solve cycles with one wave per CU (256
# environments) and sixteen (4096), and the step rate
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for v in lvs lvs_valu4 lvs_salu4 lvs_nop4 lvs_ldsr2 lvs_ldsw2; do
  V=$R/assistive_gym_amd/lib/variants/$v.so
  AGX_SOLVE_LDS_BYTES=10240 AGX_LIB=$V timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep "solve cycles" | sed "s/^lvs[a-z0-9_]*.so/$v/" | tee -a $O/cycles.txt
  AGX_SOLVE_LDS_BYTES=10240 AGX_LIB=$V timeout 300 $B > $O/bench_$v.json 2>/dev/null; line $v < $O/bench_$v.json | tee -a $O/ab.txt
done
