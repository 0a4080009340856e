#!/bin/bash
# round 4, session 18: the whole GPU suite and smoke() on the round's final tree
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04r; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^E  |passed|failed" $O/pytest_gpu.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
