#!/bin/bash
# round 4, session 3: cloth kernel A/B (round-3 barriers / LDS-only barriers / + deeper link prefetch), dressing parity, the rewritten water
# kernel (tests, bench, trace), the tightened parity tests, PMC passes over the round-4 feeding / drinking kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
for V in cloth_r3 cloth_ldsbar_pf1; do
  AGX_LIB=$R/assistive_gym_amd/lib/variants/$V.so timeout 300 python3 bench.py --task dressing --steps 30 --warmup 5 --no-cpu-baseline > $O/ab_dressing_$V.json 2> $O/ab_dressing_$V.err; cut -c1-150 $O/ab_dressing_$V.json
done
timeout 300 python3 bench.py --task dressing --steps 30 --warmup 5 --no-cpu-baseline > $O/ab_dressing_default_pf3.json 2> $O/ab_dressing_default_pf3.err; cut -c1-150 $O/ab_dressing_default_pf3.json
timeout 900 python -m pytest tests/test_gpu_dressing.py tests/test_zz_gpu_drinking.py tests/test_gpu_scratch_itch_robots.py tests/test_gpu_stretch.py tests/test_gpu_arm_manipulation.py tests/test_gpu_bed_bathing.py "tests/test_gpu_parity.py::test_noop_retest_rule_against_the_plain_solve" -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; grep -E "conditioned|VIOLENT|worst|cloth_force|oracle vs|device vs|passed|failed" $O/pytest_new.log | tail -40
timeout 300 python3 bench.py --task drinking --steps 400 > $O/bench_drinking.json 2> $O/bench_drinking.err; cut -c1-150 $O/bench_drinking.json
cd /tmp && export TMPDIR=/tmp
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_drinking -- python $R/bench.py --task drinking --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_drinking.json 2> $O/stats_drinking.err
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dressing -- python $R/bench.py --task dressing --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_dressing.json 2> $O/stats_dressing.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_feeding -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_under_rocprof_feeding.json 2> $O/stats_feeding.err
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_feeding -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof_feeding.json 2> $O/stats_unchunked_feeding.err
for T in feeding drinking; do
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/${C}_$T -- python $R/tools/pmc_workload.py $T > /dev/null 2> $O/${C}_$T.err
  done
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pmc_workload_$T -- python $R/tools/pmc_workload.py $T > /dev/null 2> $O/stats_pmc_workload_$T.err
done
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
