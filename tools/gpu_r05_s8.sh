#!/bin/bash
# round 5, session 8: the row-local sweep with scalar row headers and register impulses (-DAGX_PGS_LV=3, csrc/agx_pgs_lvs.h) against the default
# (headers in LDS, csrc/agx_pgs_lv.h): bit-for-bit states first, then the step rate over the LDS size of the solve launch, cycles per visit, parity
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/lvs.so
AGX_SOLVE_LDS_BYTES=20480 timeout 200 python tools/gpu_lv_bits.py $O/bits_lv.npz 1024 40 2>&1 | tail -1
AGX_SOLVE_LDS_BYTES=20480 AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_lvs.npz 1024 40 2>&1 | tail -1
python tools/gpu_lv_bits.py --compare $O/bits_lv.npz $O/bits_lvs.npz 2>&1 | tee $O/bits.txt; rm -f $O/bits_lv.npz $O/bits_lvs.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
timeout 300 $B > $O/bench_default.json 2>/dev/null; line default_lv_20480 < $O/bench_default.json | tee -a $O/ab.txt
for L in 9536 10240 11264 12288 13312 14336; do AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$V timeout 300 $B > $O/bench_lvs_$L.json 2>/dev/null; line lvs_lds$L < $O/bench_lvs_$L.json | tee -a $O/ab.txt; done
for L in 9536 11264 14336; do
AGX_SOLVE_LDS_BYTES=$L AGX_LIB=$V timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $O/cycles.txt
done
AGX_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "step_matches_oracle or oracle_parity_at_bench_size or episode_invariants or golden" > $O/pytest_lvs.log 2>&1; echo "lvs pytest rc=$?"; tail -3 $O/pytest_lvs.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_bench_cli.py -m gpu -q -x > $O/pytest_bench_cli.log 2>&1; echo "bench cli rc=$?"; tail -3 $O/pytest_bench_cli.log | cut -c1-300
