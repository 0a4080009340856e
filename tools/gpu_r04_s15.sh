#!/bin/bash
# round 4, session 15: SQ counters of the round-4 build and solve kernels (two PMC passes, unchunked feeding workload) and the HBM traffic passes
# (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) of the final kernels -- separate --pmc runs, no tracing flags
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04o; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc1 -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/pmc1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc2 -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/pmc2.err
for C in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${C}_feeding -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/${C}_feeding.err; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pmc_workload_feeding -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/stats_pmc_workload_feeding.err
cd $R
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in (1, 2):
    for f in glob.glob('$O/pmc%d/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k in ('agx_solve_kernel', 'agx_build_kernel', 'agx_finish_kernel'): acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
out = {k: {c: v / n[(k, c)] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open('$O/sq_counters_feeding.json', 'w'), indent=1)
for k, d in out.items(): print(k, {c: '%.4g' % v for c, v in d.items()})
PY
