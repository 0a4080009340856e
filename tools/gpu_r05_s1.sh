#!/bin/bash
# round 5, session 1: the row-local solve sweep (csrc/agx_pgs_lv.h) on hardware for the first time -- GPU suite, smoke, same-box A/B against the
# register sweep (-DAGX_PGS_LV=0), LDS budget sweep of the solve launch, kernel trace
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
timeout 300 $B > $O/bench_lv_default.json 2>$O/bench_lv_default.err; line lv_default < $O/bench_lv_default.json | tee -a $O/ab.txt
AGX_LIB=$R/assistive_gym_amd/lib/variants/reg.so timeout 300 $B > $O/bench_reg.json 2>/dev/null; line reg < $O/bench_reg.json | tee -a $O/ab.txt
for L in 12288 16384 20480; do AGX_SOLVE_LDS_BYTES=$L timeout 300 $B > $O/bench_lv_lds$L.json 2>/dev/null; line lv_lds$L < $O/bench_lv_lds$L.json | tee -a $O/ab.txt; done
timeout 300 $B > $O/bench_lv_default_2.json 2>/dev/null; line lv_default_2 < $O/bench_lv_default_2.json | tee -a $O/ab.txt
AGX_LIB=$R/assistive_gym_amd/lib/variants/reg.so timeout 300 $B > $O/bench_reg_2.json 2>/dev/null; line reg_2 < $O/bench_reg_2.json | tee -a $O/ab.txt
for C in 1 2 4; do AGX_CHUNKS=$C timeout 300 $B > $O/bench_lv_chunks$C.json 2>/dev/null; line lv_chunks$C < $O/bench_lv_chunks$C.json | tee -a $O/ab.txt; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o lv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_under_rocprof.json 2>$O/rocprof.err ); find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_feeding.csv; head -8 $O/kernel_stats_feeding.csv
( cd /tmp && AGX_CHUNKS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof1 -o lv1 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof.json 2>>$O/rocprof.err ); find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_unchunked_feeding.csv; head -6 $O/kernel_stats_unchunked_feeding.csv
rm -rf $O/prof $O/prof1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^E  |passed|failed" $O/pytest_gpu.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
