#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (all tasks), kernel-trace stats (chunked + AGX_CHUNKS=1), PMC passes.
# Every command has its own timeout.  Outputs under gpurun_out/.   tools/gpu_round.sh [quick]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
STEPS=${BENCH_STEPS:-2000}
timeout 400 python bench.py --steps $STEPS > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 300 python bench.py --task bedbathing --steps $STEPS > $O/bench_bedbathing.json 2> $O/bench_bedbathing.err; cut -c1-200 $O/bench_bedbathing.json
timeout 300 python bench.py --task scratchitch --steps 600 > $O/bench_scratchitch.json 2> $O/bench_scratchitch.err; cut -c1-200 $O/bench_scratchitch.json
timeout 300 python bench.py --task armmanipulation --steps $STEPS > $O/bench_armmanipulation.json 2> $O/bench_armmanipulation.err; cut -c1-200 $O/bench_armmanipulation.json
timeout 400 python bench.py --task dressing --steps 100 --warmup 5 > $O/bench_dressing.json 2> $O/bench_dressing.err; cut -c1-200 $O/bench_dressing.json
[ "${1:-}" = quick ] && exit 0
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof && mkdir -p $O/prof
for T in feeding bedbathing; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats_$T -- python $R/bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline > $O/prof/bench_under_rocprof_$T.json 2> $O/prof/stats_$T.err
  AGX_CHUNKS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats_unchunked_$T -- python $R/bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline > $O/prof/bench_unchunked_under_rocprof_$T.json 2> $O/prof/stats_unchunked_$T.err
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/prof/${C}_$T -- python $R/tools/pmc_workload.py $T > /dev/null 2> $O/prof/${C}_$T.err
  done
done
for T in scratchitch dressing; do
  AGX_CHUNKS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats_unchunked_$T -- python $R/bench.py --task $T --steps 10 --warmup 2 --no-cpu-baseline > $O/prof/bench_unchunked_under_rocprof_$T.json 2> $O/prof/stats_unchunked_$T.err
done
find $O/prof -name "*kernel_stats.csv" | head -40
