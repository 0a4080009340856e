#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel-trace stats, PMC passes.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof && mkdir -p $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/prof/bench_under_rocprof.json 2> $O/prof/stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof/fetch -- python $R/tools/pmc_workload.py > /dev/null 2> $O/prof/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof/write -- python $R/tools/pmc_workload.py > /dev/null 2> $O/prof/write.err
find $O/prof -name "*.csv" | head -20
