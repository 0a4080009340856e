#!/bin/bash
# round 5, session 23: the feeding variant's worklist with room for three full narrowphase passes (192 entries, arena 3820 words, build LDS 21.3 KB)
# against 166 entries (arena3592.so: an ordinary substep was flushed as 163 + 6 = four passes): bit-for-bit states, step rate, build phases
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05w; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/arena3592.so
timeout 200 python tools/gpu_lv_bits.py $O/bits_192.npz 1024 40 2>&1 | tail -1
AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_166.npz 1024 40 2>&1 | tail -1
python tools/gpu_lv_bits.py --compare $O/bits_192.npz $O/bits_166.npz 2>&1 | tee $O/bits.txt; rm -f $O/bits_*.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for r in 1 2; do
timeout 300 $B > $O/bench_192_$r.json 2>/dev/null; line worklist_192_$r < $O/bench_192_$r.json | tee -a $O/ab.txt
AGX_LIB=$V timeout 300 $B > $O/bench_166_$r.json 2>/dev/null; line worklist_166_$r < $O/bench_166_$r.json | tee -a $O/ab.txt
done
timeout 300 python tools/gpu_build_phases.py FeedingJacoVecEnv 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/build_phases_192.txt; grep -E "collide \(all\)|narrowphase" $O/build_phases_192.txt
AGX_LIB=$V timeout 300 python tools/gpu_build_phases.py FeedingJacoVecEnv 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/build_phases_166.txt; grep -E "collide \(all\)|narrowphase" $O/build_phases_166.txt
