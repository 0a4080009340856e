#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (both tasks), kernel-trace stats (chunked + AGX_CHUNKS=1), PMC passes.
# Outputs under gpurun_out/.   tools/gpu_round.sh [quick]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
STEPS=${BENCH_STEPS:-2000}
timeout 900 python bench.py --steps $STEPS > $O/bench.json 2> $O/bench.err; cat $O/bench.json
timeout 900 python bench.py --task bedbathing --steps $STEPS > $O/bench_bedbathing.json 2> $O/bench_bedbathing.err; cat $O/bench_bedbathing.json
[ "${1:-}" = quick ] && exit 0
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof && mkdir -p $O/prof
for T in feeding bedbathing; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats_$T -- python $R/bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline > $O/prof/bench_under_rocprof_$T.json 2> $O/prof/stats_$T.err
  AGX_CHUNKS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats_unchunked_$T -- python $R/bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline > $O/prof/bench_unchunked_under_rocprof_$T.json 2> $O/prof/stats_unchunked_$T.err
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/prof/${C}_$T -- python $R/tools/pmc_workload.py $T > /dev/null 2> $O/prof/${C}_$T.err
  done
done
find $O/prof -name "*.csv" | head -40
