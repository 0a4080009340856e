#!/bin/bash
# round 5, session 9 (and 10, with the impulses in LDS): the scalar-header row-local sweep with rows beyond the LDS window read from the scratch record (16 solve waves per CU at the
# default 9.5 KB): bit-for-bit states against the default build (window as large as the scene, default window, 300-pair window), step rate over
# LDS size and chunk count, cycles per visit
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/lvs.so; VC=$R/assistive_gym_amd/lib/variants/lvs_cap.so
AGX_SOLVE_LDS_BYTES=20480 timeout 200 python tools/gpu_lv_bits.py $O/bits_lv.npz 1024 40 2>&1 | tail -1
AGX_SOLVE_LDS_BYTES=20480 AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_lvs_20480.npz 1024 40 2>&1 | tail -1
AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_lvs.npz 1024 40 2>&1 | tail -1
AGX_LIB=$VC timeout 200 python tools/gpu_lv_bits.py $O/bits_lvs_cap.npz 1024 40 2>&1 | tail -1
for f in bits_lvs_20480 bits_lvs bits_lvs_cap; do python tools/gpu_lv_bits.py --compare $O/bits_lv.npz $O/$f.npz 2>&1 | tee -a $O/bits.txt; done; rm -f $O/bits_*.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
timeout 300 $B > $O/bench_default.json 2>/dev/null; line default_lv_20480 < $O/bench_default.json | tee -a $O/ab.txt
for L in 9536 10240 11264 12288; do AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$V timeout 300 $B > $O/bench_lvs_$L.json 2>/dev/null; line lvs_lds$L < $O/bench_lvs_$L.json | tee -a $O/ab.txt; done
for C in 1 2 4; do AGX_CHUNKS=$C AGX_LIB=$V timeout 300 $B > $O/bench_lvs_c$C.json 2>/dev/null; line lvs_lds9536_chunks$C < $O/bench_lvs_c$C.json | tee -a $O/ab.txt; done
AGX_LIB=$VC timeout 300 $B > $O/bench_lvs_cap.json 2>/dev/null; line lvs_window300 < $O/bench_lvs_cap.json | tee -a $O/ab.txt
for L in 9536 12288; do
AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$V timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/cycles.txt
done
AGX_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "step_matches_oracle or oracle_parity_at_bench_size or episode_invariants or golden" > $O/pytest_lvs.log 2>&1; echo "lvs pytest rc=$?"; tail -3 $O/pytest_lvs.log | cut -c1-200
