"""GPU box: cost of the device-side reset of ScratchItchPR2 (human, target, base pose search with 50 candidates per environment, collision
rejection) at 4096 environments, and what the search found:  python tools/gpu_toc_timing.py [model] [n]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
model = sys.argv[1] if len(sys.argv) > 1 else 'scratch_itch_pr2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
blob = ModelBlob.load(model)
st = Stepper(blob, n)
info = torch.zeros((n, 4), device='cuda')
st.sample_reset(1); st.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
s = torch.cuda.current_stream().cuda_stream
ev[0].record(); st.sample_reset(1001, ik_info=info, stream=s); ev[1].record()
torch.cuda.synchronize()
gi = info.cpu().numpy()
flags = st.check_collisions() if hasattr(st, 'check_collisions') else None
out = dict(model=model, envs=n, sample_reset_ms=ev[0].elapsed_time(ev[1]), start_pose_reached_frac=float(gi[:, 0].mean()), rounds_mean=float(gi[:, 1].mean()),
           rounds_max=float(gi[:, 1].max()), goals_reached_hist=np.bincount(np.clip(gi[:, 2].astype(int), 0, 4), minlength=5).tolist())
if flags is not None:
    out['still_colliding_frac'] = float((np.asarray(flags) != 0).mean())
print(json.dumps(out))
st.close()
