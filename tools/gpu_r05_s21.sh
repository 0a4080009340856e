#!/bin/bash
# round 5, session 21: scalar-cache warm-up of the next part's first header lines at the start of a part (against -DAGX_LVS_NO_PREFETCH)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/nopf.so
timeout 200 python tools/gpu_lv_bits.py $O/bits_pf.npz 1024 40 2>&1 | tail -1
AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_nopf.npz 1024 40 2>&1 | tail -1
python tools/gpu_lv_bits.py --compare $O/bits_pf.npz $O/bits_nopf.npz 2>&1 | tee $O/bits.txt; rm -f $O/bits_*.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for r in 1 2; do
timeout 300 $B > $O/bench_pf_$r.json 2>/dev/null; line warm_up_$r < $O/bench_pf_$r.json | tee -a $O/ab.txt
AGX_LIB=$V timeout 300 $B > $O/bench_nopf_$r.json 2>/dev/null; line no_warm_up_$r < $O/bench_nopf_$r.json | tee -a $O/ab.txt
done
timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep "solve cycles" | sed 's/^libagx.so/warm_up/' | tee -a $O/cycles.txt
AGX_LIB=$V timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep "solve cycles" | sed 's/^nopf.so/no_warm_up/' | tee -a $O/cycles.txt
# (rows beyond the window now wait for their own pair only, vmcnt(1): what a smaller window costs)
for L in 9536 10240; do AGX_SOLVE_LDS_BYTES=$L timeout 300 $B > $O/bench_lds$L.json 2>/dev/null; line warm_up_lds$L < $O/bench_lds$L.json | tee -a $O/ab.txt; done
timeout 600 python -m pytest tests/test_gpu_solve_variants.py -m gpu -q -x 2>&1 | tail -2
