#!/bin/bash
# round 5, session 17: 64-byte row headers for every variant (wide_all.so) against 64 bytes for the feeding variant only and the 10 words of
# rounds 1-4 for the others: the configurations whose solve kernel is the register / row-space sweep, interleaved runs
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
export TMPDIR=/tmp
V=$R/assistive_gym_amd/lib/variants/wide_all.so
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value']), j['ms_per_step'], {k[4:-7]: round(x,2) for k,x in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})"; }
for r in 1 2; do for T in bedbathing scratchitch feedingsawyer; do
  B="python bench.py --task $T --steps 300 --warmup 20 --no-cpu-baseline"
  timeout 300 $B > $O/bench_${T}_compact_$r.json 2>/dev/null; line ${T}_10_word_headers_$r < $O/bench_${T}_compact_$r.json | tee -a $O/ab.txt
  AGX_LIB=$V timeout 300 $B > $O/bench_${T}_wide_$r.json 2>/dev/null; line ${T}_64_byte_headers_$r < $O/bench_${T}_wide_$r.json | tee -a $O/ab.txt
done; done
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_feeding.json 2>/dev/null; line feeding < $O/bench_feeding.json | tee -a $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_solve_variants.py tests/test_gpu_parity.py tests/test_gpu_bed_bathing.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
