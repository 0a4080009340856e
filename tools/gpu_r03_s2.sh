#!/bin/bash
# round 3, GPU session 2: the packed solve kernel (A/B against AGX_SOLVE=old), the cloth kernel after the bank-aware link schedule
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
rm -rf $O && mkdir -p $O
cd $R
(timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/gputest.log; tail -3 $O/gputest.log
timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline > $O/ab_feeding_packed.json 2> $O/ab1.err
AGX_SOLVE=old timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline > $O/ab_feeding_old.json 2> $O/ab0.err
AGX_CHUNKS=1 timeout 200 python bench.py --task feeding --steps 200 --warmup 20 --no-cpu-baseline > $O/ab_feeding_packed_unchunked.json 2> $O/ab2.err
timeout 300 python bench.py --task dressing --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_dressing.json 2> $O/bd.err
timeout 200 python bench.py --task bedbathing --workload wiping --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_wiping.json 2> $O/bw.err
python - <<PY
import json
for f in ('ab_feeding_packed', 'ab_feeding_old', 'ab_feeding_packed_unchunked', 'bench_dressing', 'bench_wiping'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value']), j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'], j['contacts_per_substep'], j['overflow_count'])
    except Exception as e: print(f, 'failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof_feeding.json 2> $O/stats_feeding.err
AGX_CHUNKS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_feeding.json 2> $O/stats_unchunked_feeding.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/stats_dressing.err
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_dressing -- python $R/tools/pmc_workload.py dressing > /dev/null 2> $O/pmc_sq_dressing.err
for d in $O/stats_feeding $O/stats_unchunked_feeding $O/stats_dressing; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -6 $f | cut -d, -f1-6; done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('$O/pmc_sq_dressing/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'cloth' in k: acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items(): print(k, {c: '%.4g' % v for c, v in d.items()})
PY
