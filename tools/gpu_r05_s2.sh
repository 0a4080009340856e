#!/bin/bash
# round 5, session 2: what a row visit of the row-local sweep costs -- cycles per visit alone on a CU (256 environments) and sixteen to a CU (4096),
# against the register sweep, and timing builds with one piece of the visit removed each (results of those builds are meaningless)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 200 python tools/gpu_lv_cycles.py 256 1024 4096 2>&1 | grep -v Warn | tee -a $O/cycles.txt
AGX_SOLVE_LDS_BYTES=20480 timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep -v Warn | tee -a $O/cycles.txt
for v in reg full nolamw noscatter nogather nohdr noent nomem; do AGX_LIB=$R/assistive_gym_amd/lib/variants/$v.so timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep -v Warn | tee -a $O/cycles.txt; done
for v in lv reg; do
  L=$R/assistive_gym_amd/lib/libagx.so; [ $v = reg ] && L=$R/assistive_gym_amd/lib/variants/reg.so
  ( cd /tmp && AGX_LIB=$L AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof_$v.json 2>$O/rocprof_$v.err )
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_unchunked_feeding_$v.csv; head -5 $O/kernel_stats_unchunked_feeding_$v.csv; rm -rf $O/prof_$v
done
