#!/bin/bash
# round 5, session 26: the final tree -- smoke() and the driver's command
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 130 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2>$O/driver_cmd.err; python -c "
import json; j=json.loads(open('$O/driver_cmd.json').read().strip().splitlines()[-1]); print('driver cmd', round(j['value']), j['ms_per_step'], 'roofline frac', j['roofline']['frac'], 'cpu', j.get('cpu_baseline',{}).get('value'))"
