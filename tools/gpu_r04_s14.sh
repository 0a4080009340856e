#!/bin/bash
# round 4, session 14: the auxiliary reset handles attached by the base VecEnv (bed bathing: rag doll; arm manipulation: fall + rag doll) for any
# class or model name -- the device-reset tests of both tasks, ArmManipulationPR2 (two arm chains) and BedBathingPR2 through --env
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_arm_manipulation.py tests/test_gpu_bed_bathing.py tests/test_gpu_bed_bathing_robots.py -m gpu -q -k "reset or vec_env or device" > $O/pytest_reset.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_reset.log; grep -E "passed|failed|^FAILED|^E  " $O/pytest_reset.log | tail -8
timeout 400 python3 bench.py --env ArmManipulationPR2-v1 --reset device --steps 400 --no-cpu-baseline > $O/bench_armmanipulation_pr2_device_reset.json 2> $O/bench_armmanipulation_pr2_device_reset.err; cut -c1-130 $O/bench_armmanipulation_pr2_device_reset.json; tail -1 $O/bench_armmanipulation_pr2_device_reset.err
timeout 400 python3 bench.py --env BedBathingPR2-v1 --reset device --steps 400 --no-cpu-baseline > $O/bench_bedbathing_pr2_device_reset.json 2> $O/bench_bedbathing_pr2_device_reset.err; cut -c1-130 $O/bench_bedbathing_pr2_device_reset.json; tail -1 $O/bench_bedbathing_pr2_device_reset.err
