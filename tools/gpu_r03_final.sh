#!/bin/bash
# round 3, final GPU session: the whole -m gpu suite, smoke(), the default bench line, kernel traces (feeding chunked + unchunked, dressing
# unchunked), PMC passes over the dressing variant's kernels (traffic file of bench.py --task dressing)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03p}
rm -rf $O && mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/gputest.log; tail -3 $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    j = json.load(open('$O/bench_default.json')); print('default', round(j['value']), {k: round(v['value']) for k, v in j.get('configs', {}).items()}, 'cpu', j.get('cpu_baseline', {}).get('value'))
except Exception as e: print('default failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_under_rocprof_feeding.json 2> $O/s1.err
AGX_CHUNKS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_unchunked_under_rocprof_feeding.json 2> $O/s2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pmc_workload_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/s3.err
if [ -n "${AGX_FINAL_NO_PMC:-}" ]; then echo "(PMC passes skipped: AGX_FINAL_NO_PMC)"; else
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_$c.err
done
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq1_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_sq2.err
fi
for d in $O/stats_feeding $O/stats_unchunked_feeding $O/stats_pmc_workload_dressing; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -5 $f | cut -d, -f1-6; done
