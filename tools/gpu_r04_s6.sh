#!/bin/bash
# round 4, session 6: the three parity tests that were still red, the cloth kernel with 512 threads (A/B on the small-batch tool), Drinking
# unchunked (the new default), then the driver's command, the default line and the whole suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_scratch_itch_robots.py "tests/test_gpu_parity.py::test_noop_retest_rule_against_the_plain_solve" "tests/test_gpu_parity.py::test_warm_start_switch_on_the_device" -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; grep -E "conditioned|step-level|passed|failed|^FAILED|^E  " $O/pytest_new.log | tail -20
timeout 200 python3 tools/gpu_cloth_bench.py 256 5 > $O/cloth_bench_t1024.txt 2>&1; tail -1 $O/cloth_bench_t1024.txt
AGX_LIB=$R/assistive_gym_amd/lib/variants/cloth_t512.so AGX_CLOTH_BLOB=dressing_baxter_t512 timeout 200 python3 tools/gpu_cloth_bench.py 256 5 > $O/cloth_bench_t512.txt 2>&1; tail -1 $O/cloth_bench_t512.txt
timeout 200 python3 tools/gpu_cloth_bench.py 4096 2 > $O/cloth_bench_t1024_4096.txt 2>&1; tail -1 $O/cloth_bench_t1024_4096.txt
AGX_LIB=$R/assistive_gym_amd/lib/variants/cloth_t512.so AGX_CLOTH_BLOB=dressing_baxter_t512 timeout 200 python3 tools/gpu_cloth_bench.py 4096 2 > $O/cloth_bench_t512_4096.txt 2>&1; tail -1 $O/cloth_bench_t512_4096.txt
timeout 300 python3 bench.py --task drinking --steps 400 > $O/bench_drinking.json 2> $O/bench_drinking.err; cut -c1-150 $O/bench_drinking.json
timeout 300 python3 bench.py --task drinking --steps 200 --reset device --no-cpu-baseline > $O/bench_drinking_device_reset.json 2> $O/bench_drinking_device_reset.err; cut -c1-120 $O/bench_drinking_device_reset.json
for E in DrinkingSawyer-v1 DrinkingPR2-v1 DrinkingStretch-v1; do timeout 300 python3 bench.py --env $E --steps 200 --no-cpu-baseline > $O/bench_$E.json 2> $O/bench_$E.err; cut -c1-120 $O/bench_$E.json; done
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; cut -c1-160 $O/driver_cmd.json
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
