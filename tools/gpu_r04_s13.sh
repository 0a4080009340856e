#!/bin/bash
# round 4, session 13 (the round's last full run): the whole GPU suite, smoke(), kernel traces of the headline (chunked and unchunked), the
# driver's command, the default bench line, ArmManipulationPR2 with device resets (two arm chains)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^E  |passed|failed" $O/pytest_gpu.log | tail -12
mv gpurun_out/*.npz $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_feeding -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_under_rocprof_feeding.json 2> $O/stats_feeding.err
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_feeding -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof_feeding.json 2> $O/stats_unchunked_feeding.err
cd $R
find $O -name "*kernel_stats.csv" | head
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; cut -c1-130 $O/driver_cmd.json
timeout 400 python3 bench.py --env ArmManipulationPR2-v1 --reset device --steps 400 --no-cpu-baseline > $O/bench_armmanipulation_pr2_device_reset.json 2> $O/bench_armmanipulation_pr2_device_reset.err; cut -c1-130 $O/bench_armmanipulation_pr2_device_reset.json; tail -2 $O/bench_armmanipulation_pr2_device_reset.err
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-160 $O/bench_default.json
