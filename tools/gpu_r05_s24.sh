#!/bin/bash
# round 5, session 24: the sweep lists as bytes -> 174 usable worklist entries instead of 166 (wl166.so), no LDS growth: an ordinary substep in one flush of three narrowphase passes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/wl166.so
timeout 200 python tools/gpu_lv_bits.py $O/bits_174.npz 1024 40 2>&1 | tail -1
AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_166.npz 1024 40 2>&1 | tail -1
python tools/gpu_lv_bits.py --compare $O/bits_174.npz $O/bits_166.npz 2>&1 | tee $O/bits.txt; rm -f $O/bits_*.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for r in 1 2; do
timeout 300 $B > $O/bench_174_$r.json 2>/dev/null; line worklist_174_$r < $O/bench_174_$r.json | tee -a $O/ab.txt
AGX_LIB=$V timeout 300 $B > $O/bench_166_$r.json 2>/dev/null; line worklist_166_$r < $O/bench_166_$r.json | tee -a $O/ab.txt
done
timeout 300 python tools/gpu_build_phases.py FeedingJacoVecEnv 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/build_phases_174.txt; grep -E "collide \(all\)|narrowphase" $O/build_phases_174.txt
AGX_LIB=$V timeout 300 python tools/gpu_build_phases.py FeedingJacoVecEnv 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/build_phases_166.txt; grep -E "collide \(all\)|narrowphase" $O/build_phases_166.txt
