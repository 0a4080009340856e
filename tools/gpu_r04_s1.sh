#!/bin/bash
# round 4, session 1: the driver's exact command in fresh processes (twice), the same with the observation all-gather path forced on
# (1-rank RCCL group, side stream), hardware-queue settings A/B, the 2,000-step line, then the GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_1.json 2> $O/driver_cmd_1.err; cut -c1-160 $O/driver_cmd_1.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/driver_cmd_2.json 2> $O/driver_cmd_2.err; cut -c1-160 $O/driver_cmd_2.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --force-gather > $O/driver_cmd_force_gather.json 2> $O/driver_cmd_force_gather.err; cut -c1-160 $O/driver_cmd_force_gather.json
GPU_MAX_HW_QUEUES=4 timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-configs --no-cpu-baseline --force-gather > $O/hwq4_force_gather.json 2> $O/hwq4_force_gather.err; cut -c1-160 $O/hwq4_force_gather.json
GPU_MAX_HW_QUEUES=8 timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-configs --no-cpu-baseline --force-gather > $O/hwq8_force_gather.json 2> $O/hwq8_force_gather.err; cut -c1-160 $O/hwq8_force_gather.json
GPU_MAX_HW_QUEUES=4 timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $O/hwq4.json 2> $O/hwq4.err; cut -c1-160 $O/hwq4.json
GPU_MAX_HW_QUEUES=8 timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $O/hwq8.json 2> $O/hwq8.err; cut -c1-160 $O/hwq8.json
timeout 300 python3 bench.py --steps 2000 --no-configs --no-cpu-baseline > $O/bench_2000.json 2> $O/bench_2000.err; cut -c1-160 $O/bench_2000.json
timeout 800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
