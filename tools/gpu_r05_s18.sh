#!/bin/bash
# round 5, session 18: the row headers of the row-local sweep as a 32-byte table (two rows per cache line) beside a second table for the register
# sweep, against one 64-byte table (hdr64.so): bit-for-bit states, step rate (interleaved), cycles, HBM-side traffic (PMC passes)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/hdr64.so
timeout 200 python tools/gpu_lv_bits.py $O/bits_32.npz 1024 40 2>&1 | tail -1
AGX_LIB=$V timeout 200 python tools/gpu_lv_bits.py $O/bits_64.npz 1024 40 2>&1 | tail -1
python tools/gpu_lv_bits.py --compare $O/bits_32.npz $O/bits_64.npz 2>&1 | tee $O/bits.txt; rm -f $O/bits_*.npz
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for r in 1 2; do
timeout 300 $B > $O/bench_32_$r.json 2>/dev/null; line headers_32_bytes_$r < $O/bench_32_$r.json | tee -a $O/ab.txt
AGX_LIB=$V timeout 300 $B > $O/bench_64_$r.json 2>/dev/null; line headers_64_bytes_$r < $O/bench_64_$r.json | tee -a $O/ab.txt
done
AGX_CHUNKS=1 timeout 300 $B > $O/bench_32_c1.json 2>/dev/null; line headers_32_bytes_unchunked < $O/bench_32_c1.json | tee -a $O/ab.txt
AGX_CHUNKS=1 AGX_LIB=$V timeout 300 $B > $O/bench_64_c1.json 2>/dev/null; line headers_64_bytes_unchunked < $O/bench_64_c1.json | tee -a $O/ab.txt
timeout 200 python tools/gpu_lv_cycles.py 256 4096 2>&1 | grep "solve cycles" | tee -a $O/cycles.txt
cd /tmp
W="python $R/tools/pmc_workload.py feeding"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/$C -- $W > /dev/null 2> $O/$C.err
  find $O/$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/r05r_${C}_feeding.csv; rm -rf $O/$C
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $W > /dev/null 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05r_kernel_stats_pmc_workload_feeding.csv; rm -rf $O/stats
cd $R
python tools/pmc_traffic.py feeding $O/r05r_FETCH_SIZE_feeding.csv $O/r05r_WRITE_SIZE_feeding.csv $O/r05r_SQ_INSTS_VALU_feeding.csv --stats $O/r05r_kernel_stats_pmc_workload_feeding.csv --out $O/r05r_traffic_feeding.json 2>&1 | tail -2
python - <<PY
import json; j = json.load(open('$O/r05r_traffic_feeding.json'))['kernels']
for k in ('agx_build_kernel', 'agx_solve_kernel', 'agx_finish_kernel'): print(k, {a: round(b, 1) for a, b in j[k].items() if isinstance(b, float)})
PY
