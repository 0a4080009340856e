#!/bin/bash
# round 3, GPU session 4: packed solve kernel v4 (nibble headers, LDS-only look-ahead): diagnostics against AGX_SOLVE=old, A/B, trace
set -u
# (the packed kernel is an opt-in build since: AGX_LIB=assistive_gym_amd/lib/libagx_packed.so, see tools/gpu_p4_diag.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
rm -rf $O && mkdir -p $O
cd $R
timeout 200 python tools/gpu_p4_diag.py feeding_panda 24 5 > $O/diag_panda.log 2>&1; tail -8 $O/diag_panda.log
timeout 200 python tools/gpu_p4_diag.py feeding_jaco 256 6 > $O/diag_jaco.log 2>&1; tail -4 $O/diag_jaco.log
timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline > $O/ab_feeding_packed.json 2> $O/ab1.err
AGX_SOLVE=old timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline > $O/ab_feeding_old.json 2> $O/ab0.err
AGX_CHUNKS=1 timeout 200 python bench.py --task feeding --steps 200 --warmup 20 --no-cpu-baseline > $O/ab_feeding_packed_unchunked.json 2> $O/ab2.err
AGX_CHUNKS=2 timeout 200 python bench.py --task feeding --steps 200 --warmup 20 --no-cpu-baseline > $O/ab_feeding_packed_2chunks.json 2> $O/ab3.err
python - <<PY
import json
for f in ('ab_feeding_packed', 'ab_feeding_old', 'ab_feeding_packed_unchunked', 'ab_feeding_packed_2chunks'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value']), j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'], j['contacts_per_substep'], j['overflow_count'])
    except Exception as e: print(f, 'failed', e)
PY
cd /tmp && export TMPDIR=/tmp
AGX_CHUNKS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unchunked_feeding -- python $R/bench.py --task feeding --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_unchunked_under_rocprof_feeding.json 2> $O/stats_unchunked_feeding.err
for d in $O/stats_unchunked_feeding; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -6 $f | cut -d, -f1-6; done
cd $R
(timeout 700 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/gputest.log; tail -3 $O/gputest.log
