#!/bin/bash
# round 4, session 8: the whole GPU suite on the tree with the person-contact rule, the geometry-level conditioning, the relative travel bound and
# the second friction direction; same-box A/B of the travel bound (2 x 300 steps each, interleaved); config 3 with reset='device' after the
# rag doll's LDS halving; the driver's command
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^E  |passed|failed" $O/pytest_gpu.log | tail -15
mv gpurun_out/*.npz $O/ 2>/dev/null
STEPS=300 bash tools/ab_run.sh > $O/ab_travel.txt 2>&1; cat $O/ab_travel.txt
timeout 300 python3 bench.py --task bedbathing --reset device --steps 400 --no-cpu-baseline > $O/bench_bedbathing_device_reset.json 2> $O/bench_bedbathing_device_reset.err; cut -c1-160 $O/bench_bedbathing_device_reset.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; cut -c1-160 $O/driver_cmd.json
