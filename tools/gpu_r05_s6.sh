#!/bin/bash
# round 5, session 6: the row-local sweep with its visit loop in assembly (-DAGX_PGS_LV=2): parity first, then cycles per visit and the step rate
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
export TMPDIR=/tmp
for L in 9536 20480; do
AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$R/assistive_gym_amd/lib/variants/lv2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "step_matches_oracle or oracle_parity_at_bench_size or episode_invariants or golden" > $O/pytest_lv2_lds$L.log 2>&1; echo "lv2 lds=$L pytest rc=$?"; tail -3 $O/pytest_lv2_lds$L.log | cut -c1-200
done
for L in 9536 12288 16384 20480; do
AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$R/assistive_gym_amd/lib/variants/lv2.so timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/cycles.txt
done


B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }

for L in 9536 12288 16384 20480; do AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$R/assistive_gym_amd/lib/variants/lv2.so timeout 300 $B > $O/bench_lv2_$L.json 2>/dev/null; line lv2_lds$L < $O/bench_lv2_$L.json | tee -a $O/ab.txt; done
for C in 2 4 6; do AGX_CHUNKS=$C AGX_SOLVE_LDS_BYTES=20480 AGX_LIB=$R/assistive_gym_amd/lib/variants/lv2.so timeout 300 $B > $O/bench_lv2_20480_c$C.json 2>/dev/null; line lv2_lds20480_chunks$C < $O/bench_lv2_20480_c$C.json | tee -a $O/ab.txt; done
