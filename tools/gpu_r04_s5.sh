#!/bin/bash
# round 4, session 5: the parity tests with the float32 force floor / second-level conditioning, the warm-start switch on the device, the co-op
# batch adapter, Drinking after the water kernel's third pass; then the whole suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_bed_bathing.py tests/test_gpu_scratch_itch_robots.py "tests/test_gpu_parity.py::test_noop_retest_rule_against_the_plain_solve" "tests/test_gpu_parity.py::test_warm_start_switch_on_the_device" tests/test_reference_pinned.py tests/test_shim.py tests/test_zz_gpu_drinking.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; grep -E "conditioned|VIOLENT|passed|failed|^FAILED|^E  " $O/pytest_new.log | tail -40
timeout 300 python3 bench.py --task drinking --steps 400 > $O/bench_drinking.json 2> $O/bench_drinking.err; cut -c1-150 $O/bench_drinking.json
cd /tmp && export TMPDIR=/tmp
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_drinking -- python $R/bench.py --task drinking --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_drinking.json 2> $O/stats_drinking.err
head -4 $O/stats_drinking/*/*kernel_stats.csv
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
