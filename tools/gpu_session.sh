#!/bin/bash
# One parameterised runner for a `gpurun` session (replaces the per-session tools/gpu_r0N_sNN.sh scripts of rounds 3-5, which are in the git
# history up to commit 32ab8d6; profiles/README.md cites them by name).
#   gpurun --timeout 3000 -- 'bash tools/gpu_session.sh <tag> <action> [<action> ...]'
# writes everything under gpurun_out/<tag>/.  Actions (each bounded by its own timeout, a failure does not stop the next one):
#   suite            pytest -m gpu, with the conditioning tally -> pytest_gpu.log, conditioning_tally_gpu.json
#   tests:<expr>     pytest -m gpu -k <expr>
#   file:<path>      pytest -m gpu <path>
#   smoke            __graft_entry__.smoke()
#   driver           the driver's exact command: bench.py --gpus 1 --steps 20 --warmup 5 -> driver_cmd.json
#   default          bench.py with every config -> bench_default_all_configs.json
#   bench:<args>     bench.py <args> (commas for spaces) --no-cpu-baseline --no-configs -> bench_<args>.json
#   prof / prof1     rocprofv3 --kernel-trace --stats of bench.py --steps 50 (prof1: AGX_CHUNKS=1) -> kernel_stats_[unchunked_]feeding.csv
#   prof:<task>      the same for --task <task>
#   pmc:<task>       the separate --pmc passes over tools/pmc_workload.py <task>, reduced by tools/pmc_traffic.py -> traffic_<task>.json
#   rllib            tools/gpu_rllib_overhead.py -> rllib_overhead.json
#   py:<script>[,args]  python <script> args -> <script basename>.log
#   env:VAR=VAL / unset:VAR   environment for the actions that follow (e.g. env:AGX_SOLVE_LDS_BYTES=12288)
#   bits:<name>      tools/gpu_lv_bits.py: 40 steps of 1,024 FeedingJaco environments, default build against lib/variants/<name>.so, bit by bit -> bits_<name>.txt
#   share[:N]        the driver's N-rank command (default 2) with all ranks on this box's one GPU: bench.py --gpus N --backend gloo --share-gpus -> share_gpus_N.json
#   bitsenv:<name>,<VecEnv class>[,<class> ...]   the same for other tasks' environments (512 x 30 steps) -> bitsenv_<name>.txt
#   ab:<name>        AGX_LIB=assistive_gym_amd/lib/variants/<name>.so bench.py --steps 300 (x2, interleaved with the default build) -> ab_<name>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1])
    print('  %s: %.0f env-steps/s, %.3f ms/step' % (sys.argv[1].split('/')[-1], j['value'], j['ms_per_step']), {k: round(v) for k, v in j.items() if k.startswith('value_') and not isinstance(v, str)},
          'solve ms/launch %.4f' % j['roofline']['kernel_ms_per_launch'] if 'roofline' in j else '')
    for k, v in j.get('configs', {}).items(): print('     ', k, round(v['value']), v.get('contacts_per_substep'))
except Exception as e:
    print('  (no JSON line in %s: %s)' % (sys.argv[1], e))
PY
}
for A in "$@"; do
  K=${A%%:*}; V=${A#*:}; [ "$K" = "$A" ] && V=""
  echo "== $A"
  case $K in
    suite) AGX_DUMP_BENCH_STATES=$O/bench_states AGX_CONDITIONING_REPORT=$O/conditioning_tally_gpu.json timeout 2700 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; grep -E "^FAILED|^ERROR|passed|failed|oracle comparisons" $O/pytest_gpu.log | tail -16 ;;
    tests) N=$(echo "$V" | tr -c 'A-Za-z0-9' '_'); timeout 1800 python -m pytest tests -m gpu -q -rs -k "$V" > $O/pytest_$N.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_$N.log | tail -12 ;;
    file) N=$(basename "$V" .py); timeout 1800 python -m pytest "$V" -m gpu -q -rs -x > $O/pytest_$N.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|passed|failed|Error" $O/pytest_$N.log | tail -12 ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    driver) timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; line $O/driver_cmd.json ;;
    default) timeout 1200 python bench.py > $O/bench_default_all_configs.json 2>$O/bench_default.err; line $O/bench_default_all_configs.json ;;
    bench) N=$(echo "$V" | tr -c 'A-Za-z0-9=' '_')${AGX_SOLVE_LDS_BYTES:+_lds$AGX_SOLVE_LDS_BYTES}; timeout 900 python bench.py $(echo "$V" | tr ',' ' ') --no-cpu-baseline --no-configs > $O/bench_$N.json 2>$O/bench_$N.err; line $O/bench_$N.json ;;
    prof|prof1) T=${V:-feeding}; S=""; [ $K = prof1 ] && S="unchunked_"
      if [ $K = prof1 ]; then export AGX_CHUNKS=1; fi
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tmp -- python $R/bench.py --task $T --steps 50 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_${S}under_rocprof_$T.json 2>$O/rocprof_$T.err )
      [ $K = prof1 ] && unset AGX_CHUNKS
      find $O/prof_tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_${S}$T.csv; head -6 $O/kernel_stats_${S}$T.csv; rm -rf $O/prof_tmp ;;
    pmc) T=${V:-feeding}; mkdir -p $O/pmc
      for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
        N=$(echo $C | tr ' ' '+'); ( cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $O/pmc_tmp -- python $R/tools/pmc_workload.py $T > /dev/null 2>>$O/pmc.err )
        find $O/pmc_tmp -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc/${N}_$T.csv; rm -rf $O/pmc_tmp
      done
      python tools/pmc_traffic.py $T $O/pmc/FETCH_SIZE_$T.csv $O/pmc/WRITE_SIZE_$T.csv $O/pmc/SQ_INSTS_VALU_$T.csv --more $O/pmc/SQ_INSTS_SALU+SQ_INSTS_LDS+SQ_INSTS_SMEM_$T.csv $O/pmc/SQ_WAVE_CYCLES+SQ_WAIT_ANY+SQ_WAIT_INST_ANY+SQ_ACTIVE_INST_ANY_$T.csv $O/pmc/SQ_LDS_BANK_CONFLICT+SQ_LDS_IDX_ACTIVE+SQ_ACTIVE_INST_LDS_$T.csv --out $O/traffic_$T.json 2>>$O/pmc.err; head -c 1500 $O/traffic_$T.json ;;
    rllib) timeout 400 python tools/gpu_rllib_overhead.py 2>$O/rllib.err | tail -1 | tee $O/rllib_overhead.json ;;
    py) S=${V%%,*}; AR=$(echo "${V#*,}" | tr ',' ' '); [ "$S" = "$V" ] && AR=""; timeout 1500 python $S $AR > $O/$(basename $S .py).log 2>&1; echo "rc=$?"; tail -12 $O/$(basename $S .py).log ;;
    ab) : > $O/ab_$V.txt
      for rep in 1 2; do for L in default $V; do
        if [ $L = default ]; then unset AGX_LIB; else export AGX_LIB=$R/assistive_gym_amd/lib/variants/$V.so; fi
        timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(j['value']), round(j['ms_per_step'], 3), {k.split('_')[1]: round(v, 2) for k, v in j['roofline']['kernels_ms_per_step_summed_over_overlapping_launches'].items()})" | tee -a $O/ab_$V.txt
      done; done; unset AGX_LIB ;;
    bitsenv) for E in $(echo "${V#*,}" | tr ',' ' '); do L=${V%%,*}
        AGX_BITS_ENV=$E timeout 600 python tools/gpu_lv_bits.py /tmp/be_default.npz 512 30 > /dev/null 2>$O/bitsenv_$E.err; AGX_BITS_ENV=$E AGX_LIB=$R/assistive_gym_amd/lib/variants/$L.so timeout 600 python tools/gpu_lv_bits.py /tmp/be_$L.npz 512 30 > /dev/null 2>>$O/bitsenv_$E.err
        echo "$E: $(python tools/gpu_lv_bits.py --compare /tmp/be_default.npz /tmp/be_$L.npz | tail -1)" | tee -a $O/bitsenv_$L.txt; done ;;
    bits) timeout 600 python tools/gpu_lv_bits.py /tmp/bits_default.npz 1024 40 > /dev/null 2>$O/bits_$V.err; AGX_LIB=$R/assistive_gym_amd/lib/variants/$V.so timeout 600 python tools/gpu_lv_bits.py /tmp/bits_$V.npz 1024 40 > /dev/null 2>>$O/bits_$V.err
      python tools/gpu_lv_bits.py --compare /tmp/bits_default.npz /tmp/bits_$V.npz | tail -2 | tee $O/bits_$V.txt ;;
    share) N=${V:-2}; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 50 --warmup 5 --backend gloo --share-gpus --no-cpu-baseline > $O/share_gpus_$N.json 2>$O/share_gpus_$N.err; echo "rc=$?"; line $O/share_gpus_$N.json; tail -3 $O/share_gpus_$N.err ;;
    env) export "$V"; echo "exported $V" ;;
    unset) unset "$V" ;;
    *) echo "unknown action $A" ;;
  esac
done
