#!/bin/bash
# round 5, session 5: GPU suite after the group flag was kept off compound tools (spoon, cup), the early-done count, bench lines, RLlib adapter cost
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python tools/gpu_early_done_cases.py FeedingJacoVecEnv 4 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/early.txt
timeout 300 python tools/gpu_early_done_cases.py FeedingSawyerVecEnv 4 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/early.txt
AGX_CONDITIONING_REPORT=$O/conditioning_tally_gpu.json timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|passed|failed|oracle comparisons" $O/pytest_gpu.log | tail -14
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for rep in 1 2; do timeout 300 $B > $O/bench_default_$rep.json 2>/dev/null; line default_$rep < $O/bench_default_$rep.json | tee -a $O/ab.txt; done
timeout 300 $B --force-gather > $O/bench_force_gather_abi.json 2>$O/fg.err; line force_gather_abi < $O/bench_force_gather_abi.json | tee -a $O/ab.txt; grep -o '"gather": "[^"]*"' $O/bench_force_gather_abi.json | tee -a $O/ab.txt
timeout 300 $B --force-gather --gather torch > $O/bench_force_gather_torch.json 2>>$O/fg.err; line force_gather_torch < $O/bench_force_gather_torch.json | tee -a $O/ab.txt
for t in bedbathing scratchitch; do timeout 300 python bench.py --task $t --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_$t.json 2>/dev/null; line $t < $O/bench_$t.json | tee -a $O/ab.txt; done
timeout 300 python bench.py --task bedbathing --workload wiping --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_wiping.json 2>/dev/null; line wiping < $O/bench_wiping.json | tee -a $O/ab.txt
timeout 300 python tools/gpu_rllib_overhead.py 2>/dev/null | tail -1 | tee $O/rllib_overhead.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
