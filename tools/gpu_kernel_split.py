"""GPU box: per-kernel launch durations (agx_step_timed, unchunked) of a task with a blob parameter changed:
   AGX_CHUNKS=1 python tools/gpu_kernel_split.py BedBathingSawyerVecEnv NITER=0 NITER=50"""
import os, sys
os.environ.setdefault('AGX_CHUNKS', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd import vec_env
cls = sys.argv[1]
n = 4096
pool = None
for arg in sys.argv[2:]:
    name, val = arg.split('=')
    base = getattr(vec_env, cls)(1, pool_size=1).blob if False else None
    from assistive_gym_amd.blob import ModelBlob
    b0 = ModelBlob.load(getattr(vec_env, cls).model)
    if getattr(vec_env, cls).coop:
        b0 = b0.coop()
    env = getattr(vec_env, cls)(n, pool_size=64, seed=1001, blob=b0.set_param(name, float(val)))
    if pool is not None:
        env.pool_host, env.pool = pool, torch.from_numpy(pool).cuda()
    env.reset()
    pool = env.pool_host
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    for k in range(10):
        env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
    ms = np.zeros(3); cnt = np.zeros(3)
    for k in range(20):
        a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
        m, c = env.stepper.step_timed(a, env.obs, env.reward, env.done, env.info, torch.cuda.current_stream().cuda_stream)
        ms += m; cnt += c
    print(cls, arg, 'ms per launch: build %.3f solve %.3f finish %.3f' % tuple(ms / np.maximum(cnt, 1)))
    env.close()
