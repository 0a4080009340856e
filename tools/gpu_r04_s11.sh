#!/bin/bash
# round 4, session 11: the persistent manifold on the device; the arm-manipulation tests after the violent-start rule; same-box A/B of the default
# path with and without the manifold stage compiled into the build kernel (2 x 300 steps each, interleaved)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_persistent_manifold_on_the_device" "tests/test_gpu_arm_manipulation.py::test_other_single_arm_robots" "tests/test_gpu_parity.py::test_step_matches_oracle" -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; grep -E "VIOLENT|conditioned|passed|failed|^FAILED|^E  " $O/pytest_new.log | tail -20
STEPS=300 bash tools/ab_run.sh > $O/ab_manifold.txt 2>&1; grep -v amdgpu $O/ab_manifold.txt
