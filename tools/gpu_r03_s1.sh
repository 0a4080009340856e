#!/bin/bash
# round 3, GPU session 1: GPU suite after the PGS re-test rule, A/B of the rule, the default bench line with all single-GPU configs,
# kernel trace of the headline, counters of the cloth kernel.  Outputs under gpurun_out/r03b/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
rm -rf $O && mkdir -p $O
cd $R
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/gputest.log; tail -2 $O/gputest.log
timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline > $O/ab_feeding_retest5.json 2> $O/ab1.err
timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline --param NOOP_RETEST=0 > $O/ab_feeding_retest0.json 2> $O/ab0.err
python - <<PY
import json
for f in ('ab_feeding_retest5', 'ab_feeding_retest0'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value']), j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'], j['contacts_per_substep'], j['overflow_count'])
    except Exception as e: print(f, 'failed', e)
PY
timeout 600 python bench.py --steps 600 --warmup 30 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof_feeding.json 2> $O/stats_feeding.err
AGX_CHUNKS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_feeding.json 2> $O/stats_unchunked_feeding.err
# the cloth kernel: kernel trace + counters (separate passes; no trace domains together with --pmc)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/stats_dressing.err
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_${C}_dressing.err
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq1_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_sq1_dressing.err
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq2_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_sq2_dressing.err
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
for d in $O/stats_feeding $O/stats_unchunked_feeding $O/stats_dressing; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -8 $f | cut -d, -f1-6; done
