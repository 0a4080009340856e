#!/bin/bash
# round 3, GPU session 9: DressingBaxter with reset='device' (sampling + garment + 50-step settle on the GPU at every episode boundary)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03o}
rm -rf $O && mkdir -p $O
cd $R
timeout 500 python bench.py --task dressing --reset device --steps 410 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_dressing_device_reset.json 2> $O/e1.err
timeout 500 python bench.py --task dressing --steps 410 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_dressing_pool.json 2> $O/e2.err
python - <<PY
import json
for f in ('bench_dressing_device_reset', 'bench_dressing_pool'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value']), 'ms/step %.3f' % j['ms_per_step'], j['config']['reset'])
    except Exception as e: print(f, 'failed', e)
PY
python - <<PY > $O/dressing_reset_timing.json
import json, sys, time
sys.path.insert(0, '$R')
import torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
b = ModelBlob.load('dressing_baxter'); n = 4096
st = Stepper(b, n); st.reset(None, None, 1, settle_substeps=1); st.synchronize()
info = torch.zeros((n, 4), device='cuda')
t0 = time.time(); st.sample_reset(1001, ik_info=info); st.synchronize(); t1 = time.time()
st.reset(None, None, 2002, settle_substeps=50); st.synchronize(); t2 = time.time()
gi = info.cpu().numpy()
print(json.dumps(dict(envs=n, sample_reset_ms=(t1 - t0) * 1e3, whole_reset_incl_50_step_settle_ms=(t2 - t1) * 1e3, start_pose_reached_frac=float(gi[:, 0].mean()),
                      rounds_mean=float(gi[:, 1].mean()), goals_reached_mean=float(gi[:, 2].mean()))))
PY
cat $O/dressing_reset_timing.json
