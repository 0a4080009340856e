#!/bin/bash
# SQ wave-cycle breakdown of the three kernels (two PMC passes, unchunked workload).  Output: gpurun_out/sq/*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq
T=${1:-feeding}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -- python $R/tools/pmc_workload.py $T > /dev/null 2> $O/p1.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $O/p2 -- python $R/tools/pmc_workload.py $T > /dev/null 2> $O/p2.err
python - <<PY
import csv, glob, collections
for p in ('p1', 'p2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    for k, d in acc.items():
        print(p, k, {c: '%.4g' % v for c, v in d.items()})
PY
