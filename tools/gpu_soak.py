"""Soak run (GPU box): n envs x N steps with auto-reset, checks every 50 steps that all outputs are finite and
reports return statistics per episode (random policy).  python tools/gpu_soak.py [steps] [pool|device|host] [VecEnv class] [n_envs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from assistive_gym_amd import vec_env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
mode = sys.argv[2] if len(sys.argv) > 2 else 'pool'
cls = sys.argv[3] if len(sys.argv) > 3 else 'FeedingJacoVecEnv'
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
env = getattr(vec_env, cls)(n, pool_size=256 if n >= 1024 else 32, seed=7, reset=mode)
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
ret = torch.zeros(n, device='cuda'); ep_returns = []
bad = 0
t0 = time.time()
for k in range(N):
    a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
    obs, rew, done, info = env.step(a)
    ret += rew
    if bool(done.any()):
        ep_returns.append(ret[done.bool()].clone()); ret[done.bool()] = 0
    if k % 50 == 49:
        bad += int((~torch.isfinite(obs)).sum()) + int((~torch.isfinite(rew)).sum()) + int((~torch.isfinite(info)).sum())
torch.cuda.synchronize()
r = torch.cat(ep_returns)
print('reset mode', mode, 'steps', N, 'episodes', len(r), 'non-finite values', bad, 'return mean %.2f std %.2f min %.2f max %.2f' % (r.mean(), r.std(), r.min(), r.max()),
      'max |obs| %.2f' % float(obs.abs().max()), 'task_success mean %.3f' % float(info[:, 1].mean()), 'env-steps/s %.0f' % (n * N / (time.time() - t0)))
