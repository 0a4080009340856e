"""GPU box: FeedingJaco states after a short rollout, for a bit-for-bit comparison of two builds of the solve path that do the same arithmetic
(e.g. the row-local sweep with its headers in LDS, csrc/agx_pgs_lv.h, and with scalar headers, csrc/agx_pgs_lvs.h).
usage: AGX_LIB=<build> python tools/gpu_lv_bits.py out.npz [n_envs] [steps];  python tools/gpu_lv_bits.py --compare a.npz b.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if sys.argv[1] == '--compare':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == 'f' else np.array_equal(a[k], b[k])
        if not same:
            x, y = a[k].reshape(a[k].shape[0], -1), b[k].reshape(b[k].shape[0], -1)
            rows = np.where((x.view(np.uint32) != y.view(np.uint32)).any(axis=1))[0] if x.dtype == np.float32 else np.where((x != y).any(axis=1))[0]
            print('%s: %d of %d environments differ (first %s)' % (k, len(rows), x.shape[0], rows[:8])); bad += 1
            if a[k].ndim >= 2:            # which trailing columns, and by how much
                xs, ys = a[k].reshape(-1, a[k].shape[-1]), b[k].reshape(-1, b[k].shape[-1])
                cols = np.where((xs != ys).any(axis=0))[0]
                print('   columns', cols[:16], 'max |difference| per column', [float(np.nanmax(np.abs(xs[:, c].astype(np.float64) - ys[:, c]))) for c in cols[:16]], 'shape', a[k].shape)
    print('IDENTICAL' if not bad else 'DIFFERENT', sys.argv[2], sys.argv[3])
    sys.exit(1 if bad else 0)
import torch
from assistive_gym_amd import vec_env
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
blob = None
if os.environ.get('AGX_BITS_PARAM'):            # e.g. SOLVE_WIDE=0: the narrow sweep inside the default build (same-process A/B switch of the blob)
    from assistive_gym_amd.blob import ModelBlob
    blob = ModelBlob.load('feeding_jaco')
    for kv in os.environ['AGX_BITS_PARAM'].split(','):
        k, v = kv.split('='); blob = blob.set_param(k, float(v))
env = getattr(vec_env, os.environ.get('AGX_BITS_ENV', 'FeedingJacoVecEnv'))(n, pool_size=64, seed=1001, **({'blob': blob} if blob is not None else {}))      # AGX_BITS_ENV: another task's VecEnv class
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(1)
rew = []
for k in range(steps):
    env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
    rew.append(env.reward.clone())
torch.cuda.synchronize()
np.savez(sys.argv[1], state=env.stepper.get_state().view(np.uint32), obs=env.obs.cpu().numpy(),
         reward=torch.stack(rew).cpu().numpy(), info=env.info.cpu().numpy())
print('wrote', sys.argv[1], os.environ.get('AGX_LIB', 'libagx.so'), os.environ.get('AGX_SOLVE_LDS_BYTES', 'default'), os.environ.get('AGX_BITS_PARAM', ''))
