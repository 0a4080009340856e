#!/bin/bash
# round 5, session 4: which environments end early under the new group flag (states for emulator replay); timing ablations of the register sweep
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python tools/gpu_early_done_cases.py FeedingJacoVecEnv 16 2>&1 | grep -v Warn | tee -a $O/early.txt
timeout 300 python tools/gpu_early_done_cases.py FeedingSawyerVecEnv 8 2>&1 | grep -v Warn | tee -a $O/early.txt
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for rep in 1 2; do
timeout 300 $B > $O/bench_default_$rep.json 2>/dev/null; line default_$rep < $O/bench_default_$rep.json | tee -a $O/ab.txt
for v in nolds noaddr norl3; do AGX_LIB=$R/assistive_gym_amd/lib/variants/$v.so timeout 300 $B > $O/bench_${v}_${rep}.json 2>/dev/null; line ${v}_$rep < $O/bench_${v}_${rep}.json | tee -a $O/ab.txt; done
done
