#!/bin/bash
# round 4, session 2: Drinking as a product path (tests, bench line, kernel trace), the non-finite guard's replacement path, the Stretch on its
# centre-of-mass base frame, then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_zz_gpu_drinking.py tests/test_gpu_stretch.py "tests/test_gpu_parity.py::test_nonfinite_environment_is_flagged_masked_and_replaced" -m gpu -q -x > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log; tail -5 $O/pytest_new.log
timeout 300 python3 bench.py --task drinking --steps 400 > $O/bench_drinking.json 2> $O/bench_drinking.err; cut -c1-200 $O/bench_drinking.json; tail -2 $O/bench_drinking.err
for E in DrinkingSawyer-v1 DrinkingPR2-v1 DrinkingStretch-v1; do timeout 300 python3 bench.py --env $E --steps 200 --no-cpu-baseline > $O/bench_$E.json 2> $O/bench_$E.err; cut -c1-120 $O/bench_$E.json; tail -1 $O/bench_$E.err; done
timeout 300 python3 bench.py --task drinking --steps 200 --reset device --no-cpu-baseline > $O/bench_drinking_device_reset.json 2> $O/bench_drinking_device_reset.err; cut -c1-120 $O/bench_drinking_device_reset.json
cd /tmp && export TMPDIR=/tmp
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_drinking -- python $R/bench.py --task drinking --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_drinking.json 2> $O/stats_drinking.err
find $O/stats_drinking -name "*kernel_stats.csv" | head -3
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
