#!/bin/bash
# round 4, session 22 (the round's last GPU seconds): same-box A/B of the AABB table without its travel word (worklist of 200 instead of 182 entries:
# one flush per FeedingJaco substep instead of two; exact) against the layout of session 18
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
STEPS=150 bash tools/ab_run.sh > $O/ab_worklist.txt 2>&1; grep -v amdgpu $O/ab_worklist.txt
timeout 60 python -m pytest "tests/test_gpu_parity.py::test_step_matches_oracle" "tests/test_gpu_parity.py::test_debug_internals_match_oracle" -m gpu -q 2>&1 | tail -2
