#!/bin/bash
# round 4, session 9: ArmManipulationEnv.reset on the device (three models in a row), the two tests that were red in session 8, arm manipulation
# with pool / device resets
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_arm_manipulation.py "tests/test_gpu_parity.py::test_noop_retest_rule_against_the_plain_solve" "tests/test_gpu_parity.py::test_second_friction_direction_on_the_device" tests/test_reset_generator.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; grep -E "VIOLENT|conditioned|passed|failed|^FAILED|^E  " $O/pytest_new.log | tail -25
timeout 400 python3 bench.py --task armmanipulation --steps 400 --no-cpu-baseline > $O/bench_armmanipulation_pool.json 2> $O/bench_armmanipulation_pool.err; cut -c1-140 $O/bench_armmanipulation_pool.json; tail -2 $O/bench_armmanipulation_pool.err
timeout 400 python3 bench.py --task armmanipulation --reset device --steps 400 --no-cpu-baseline > $O/bench_armmanipulation_device_reset.json 2> $O/bench_armmanipulation_device_reset.err; cut -c1-140 $O/bench_armmanipulation_device_reset.json; tail -2 $O/bench_armmanipulation_device_reset.err
