#!/bin/bash
# round 5, session 16: PMC passes over the round-5 feeding kernels (the scalar-header row-local sweep in agx_solve_kernel): HBM traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes), instruction counts and SQ cycle counters, the kernel trace of the same workload
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
W="python $R/tools/pmc_workload.py feeding"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/$C -- $W > /dev/null 2> $O/$C.err
  find $O/$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/r05p_${C}_feeding.csv; rm -rf $O/$C
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $O/pmc1 -- $W > /dev/null 2> $O/pmc1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc2 -- $W > /dev/null 2> $O/pmc2.err
for p in 1 2; do find $O/pmc$p -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/r05p_sq_pass${p}_feeding.csv; rm -rf $O/pmc$p; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $W > /dev/null 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05p_kernel_stats_pmc_workload_feeding.csv; rm -rf $O/stats
cd $R
python tools/pmc_traffic.py feeding $O/r05p_FETCH_SIZE_feeding.csv $O/r05p_WRITE_SIZE_feeding.csv $O/r05p_SQ_INSTS_VALU_feeding.csv --more $O/r05p_sq_pass1_feeding.csv $O/r05p_sq_pass2_feeding.csv --stats $O/r05p_kernel_stats_pmc_workload_feeding.csv --out $O/r05_traffic_feeding.json 2>&1 | tail -5
python - <<PY
import json; j = json.load(open('$O/r05_traffic_feeding.json')); print(json.dumps(j, indent=1)[:3000])
PY
