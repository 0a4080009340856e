#!/bin/bash
# round 3: quick A/B of the packed solve kernel against the one-wave-per-environment kernel (AGX_SOLVE=old), feeding headline workload
set -u
# (the packed kernel is an opt-in build since: AGX_LIB=assistive_gym_amd/lib/libagx_packed.so, see tools/gpu_p4_diag.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03f}
rm -rf $O && mkdir -p $O
cd $R
timeout 200 python tools/gpu_p4_diag.py feeding_jaco 256 6 > $O/diag_jaco.log 2>&1; tail -4 $O/diag_jaco.log
timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline --no-configs > $O/ab_feeding_packed.json 2> $O/ab1.err
AGX_SOLVE=old timeout 200 python bench.py --task feeding --steps 400 --warmup 20 --no-cpu-baseline --no-configs > $O/ab_feeding_old.json 2> $O/ab0.err
AGX_CHUNKS=1 timeout 200 python bench.py --task feeding --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $O/ab_feeding_packed_unchunked.json 2> $O/ab2.err
python - <<PY
import json
for f in ('ab_feeding_packed', 'ab_feeding_old', 'ab_feeding_packed_unchunked'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value']), j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'], j['contacts_per_substep'], j['overflow_count'])
    except Exception as e: print(f, 'failed', e)
PY
