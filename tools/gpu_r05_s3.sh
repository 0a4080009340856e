#!/bin/bash
# round 5, session 3: the GPU suite with the new group flag (bit 6), the bench-size parity tests of configs 3 / 4 / 5, the rehearsal dumps through the
# C ABI, the conditioning tally; the default bench line; timing ablations of the register sweep's row-ahead address arithmetic
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
export TMPDIR=/tmp
AGX_CONDITIONING_REPORT=$O/conditioning_tally_gpu.json timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^E  |passed|failed|oracle comparisons|bench-size parity|contacts per env step|cloth_force_sum" $O/pytest_gpu.log | tail -30
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for rep in 1 2; do
timeout 300 $B > $O/bench_default_$rep.json 2>/dev/null; line default_$rep < $O/bench_default_$rep.json | tee -a $O/ab.txt
for v in nolds noaddr norl3; do AGX_LIB=$R/assistive_gym_amd/lib/variants/$v.so timeout 300 $B > $O/bench_${v}_${rep}.json 2>/dev/null; line ${v}_$rep < $O/bench_${v}_${rep}.json | tee -a $O/ab.txt; done
done
for t in bedbathing scratchitch; do timeout 300 python bench.py --task $t --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_$t.json 2>/dev/null; line $t < $O/bench_$t.json | tee -a $O/ab.txt; done
timeout 300 python bench.py --task bedbathing --workload wiping --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_wiping.json 2>/dev/null; line wiping < $O/bench_wiping.json | tee -a $O/ab.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
