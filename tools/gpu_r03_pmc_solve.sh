#!/bin/bash
# round 3: SQ counters of the two solve kernels (packed agx_solve4_kernel / AGX_SOLVE=old agx_solve_kernel) on the unchunked feeding workload
set -u
# (the packed kernel is an opt-in build since: AGX_LIB=assistive_gym_amd/lib/libagx_packed.so, see tools/gpu_p4_diag.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03h}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in new old; do
  AGX_SOLVE=$mode timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc1_$mode -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/pmc1_$mode.err
  AGX_SOLVE=$mode timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc2_$mode -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/pmc2_$mode.err
  AGX_SOLVE=$mode timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_IFETCH SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT --output-format csv -d $O/pmc3_$mode -- python $R/tools/pmc_workload.py feeding > /dev/null 2> $O/pmc3_$mode.err
done
python - <<PY
import csv, glob, collections
for mode in ('new', 'old'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for p in (1, 2, 3):
        for f in glob.glob('$O/pmc%d_%s/**/*counter_collection.csv' % (p, mode), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].split('(')[0]
                if 'solve' in k: acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in acc.items(): print(mode, k, {c: '%.4g' % (v / n[(k, c)]) for c, v in d.items()})
PY
