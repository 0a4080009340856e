#!/bin/bash
# round 5, session 27: the whole-batch gather forced on one GPU (1-rank communicator of the C ABI / torch.distributed) with the final kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05zz; mkdir -p $O; cd $R
export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs --force-gather"
timeout 70 $B > $O/bench_force_gather_abi.json 2>$O/abi.err; python -c "
import json; j=json.loads(open('$O/bench_force_gather_abi.json').read().strip().splitlines()[-1]); print('force-gather', j['config']['gather'], round(j['value']), j['ms_per_step'])"
timeout 70 $B --gather torch > $O/bench_force_gather_torch.json 2>$O/torch.err; python -c "
import json; j=json.loads(open('$O/bench_force_gather_torch.json').read().strip().splitlines()[-1]); print('force-gather', j['config']['gather'], round(j['value']), j['ms_per_step'])"
