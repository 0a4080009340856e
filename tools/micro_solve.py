"""Microbenchmark (GPU box): kernel times of ONE substep from identical valid states (FRAME_SKIP = 1), for
timing variants of the solve kernel whose results are not meaningful (ablation builds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.host.reset import make_states
blob = ModelBlob.load(); b1 = blob.set_param('FRAME_SKIP', 1)
n = 4096
cache = os.path.join(ROOT, 'gpurun_out', 'micro_states.npy')     # settled once, by the first (unmodified) build
if os.path.exists(cache):
    base = np.load(cache)
else:
    base, _ = make_states(blob, 64, seed=11)
    pre = Stepper(blob, 64); pre.set_state(base); pre.settle(25); pre.synchronize(); base = pre.get_state(); pre.close()
    os.makedirs(os.path.dirname(cache), exist_ok=True); np.save(cache, base)
states = base[np.arange(n) % 64]
st = Stepper(b1, n)
dev = torch.device('cuda', 0)
act = torch.zeros((n, 7), device=dev); obs = torch.zeros((n, 25), device=dev); rew = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, 8), device=dev)
ms = np.zeros(3)
for k in range(6):
    st.set_state(states)
    t, cnt = st.step_timed(act, obs, rew, done, info)
    if k > 0: ms += np.array(t) / np.maximum(np.array(cnt), 1)
print(os.environ.get('AGX_LIB', 'default').split('/')[-1], 'build %.3f solve %.3f finish %.3f ms per substep' % tuple(ms / 5))
