#!/bin/bash
# round 4, session 4: the whole GPU suite on blob v11 (NOOP_PEN, primitive margins, water kernel with batched plane loads), the driver's command,
# the 2,000-step line with every config, Drinking (bench + trace), the RLlib adapters' host overhead
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "conditioned|VIOLENT|passed|failed|^FAILED|free-running|NOOP_RETEST 5|cloth_force|oracle vs itself" $O/pytest_gpu.log | tail -30
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; cut -c1-160 $O/driver_cmd.json
timeout 600 python3 bench.py > $O/bench_default_all_configs.json 2> $O/bench_default_all_configs.err; cut -c1-160 $O/bench_default_all_configs.json
timeout 300 python3 bench.py --task drinking --steps 400 > $O/bench_drinking.json 2> $O/bench_drinking.err; cut -c1-160 $O/bench_drinking.json
timeout 300 python3 tools/gpu_rllib_overhead.py > $O/rllib_overhead.json 2> $O/rllib_overhead.err; cat $O/rllib_overhead.json
cd /tmp && export TMPDIR=/tmp
AGX_CHUNKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_drinking -- python $R/bench.py --task drinking --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_drinking.json 2> $O/stats_drinking.err
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/${C}_drinking -- python $R/tools/pmc_workload.py drinking > /dev/null 2> $O/${C}_drinking.err
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pmc_workload_drinking -- python $R/tools/pmc_workload.py drinking > /dev/null 2> $O/stats_pmc_workload_drinking.err
find $O -name "*kernel_stats.csv" | head
