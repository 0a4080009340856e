#!/bin/bash
# round 5, session 15: narrowphase passes in the order of the effort a pair took last time (csrc/agx_collide.h) against worklist order
# (-DAGX_NARROWPHASE_BY_EFFORT=0): bit-for-bit states, step rate (interleaved runs), build-kernel phase cycles
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05o; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/np_order.so
timeout 200 python tools/gpu_lv_bits.py $O/bits_effort.npz 1024 40 2>&1 | tail -1
AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_worklist.npz 1024 40 2>&1 | tail -1
python tools/gpu_lv_bits.py --compare $O/bits_effort.npz $O/bits_worklist.npz 2>&1 | tee $O/bits.txt; rm -f $O/bits_*.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for r in 1 2; do
timeout 300 $B > $O/bench_effort_$r.json 2>/dev/null; line by_effort_$r < $O/bench_effort_$r.json | tee -a $O/ab.txt
AGX_LIB=$V timeout 300 $B > $O/bench_worklist_$r.json 2>/dev/null; line worklist_order_$r < $O/bench_worklist_$r.json | tee -a $O/ab.txt
done
timeout 300 python tools/gpu_build_phases.py FeedingJacoVecEnv 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/build_phases_by_effort.txt; grep -E "collide|narrowphase" $O/build_phases_by_effort.txt
AGX_LIB=$V timeout 300 python tools/gpu_build_phases.py FeedingJacoVecEnv 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/build_phases_worklist_order.txt; grep -E "collide|narrowphase" $O/build_phases_worklist_order.txt
