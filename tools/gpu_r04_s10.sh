#!/bin/bash
# round 4, session 10: the whole GPU suite on the tree with the arm-manipulation device resets (all five robots), smoke(), the default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^E  |passed|failed" $O/pytest_gpu.log | tail -15
mv gpurun_out/*.npz $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json; tail -2 $O/bench_default.err
