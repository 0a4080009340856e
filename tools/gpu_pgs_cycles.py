"""GPU box: shader-clock cycles of the PGS of the first substep per environment (debug record) against its row count."""
import os, sys
os.environ.setdefault('AGX_CHUNKS', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd import vec_env
cls = sys.argv[1] if len(sys.argv) > 1 else 'BedBathingSawyerVecEnv'
n = 4096
env = getattr(vec_env, cls)(n, pool_size=64, seed=1001)
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(1)
for k in range(30):
    env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
lay = env.stepper.debug_layout()
dbg = torch.zeros((n, lay[0]), device='cuda')
a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
env.stepper.step_dev(a, env.obs, env.reward, env.done, env.info, torch.cuda.current_stream().cuda_stream, debug=dbg)
torch.cuda.synchronize()
D = dbg.cpu().numpy()
rows, cyc, cyc_a = D[:, 1], D[:, lay[6] + 5], D[:, lay[6] + 7]
for lo, hi in ((0, 20), (20, 30), (30, 40), (40, 56), (56, 64), (64, 200)):
    m = (rows > lo) & (rows <= hi)
    if m.any():
        print('rows (%d, %d]: %5d envs, pgs cycles median %.0f max %.0f, cycles per row visit (50 sweeps) %.0f; row-space set-up (dense J, A = J B^T) median %.0f = %.0f %%' % (lo, hi, m.sum(), np.median(cyc[m]), cyc[m].max(), np.median(cyc[m] / (50 * rows[m])), np.median(cyc_a[m]), 100 * np.median(cyc_a[m] / np.maximum(cyc[m], 1))))
