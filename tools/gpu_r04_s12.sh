#!/bin/bash
# round 4, session 12: the manifold as a second build kernel (default kernel untouched), the split-impulse threshold, the arm-manipulation tests
# after the violent-start rule; a 300-step line of the default path as a check against session 11's A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_persistent_manifold_on_the_device" "tests/test_gpu_parity.py::test_split_impulse_threshold_on_the_device" "tests/test_gpu_arm_manipulation.py::test_other_single_arm_robots" "tests/test_gpu_parity.py::test_step_matches_oracle" -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; grep -E "VIOLENT|conditioned|passed|failed|^FAILED|^E  " $O/pytest_new.log | tail -20
for r in 1 2; do timeout 300 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_300_$r.json 2> $O/bench_300_$r.err; cut -c1-110 $O/bench_300_$r.json; done
