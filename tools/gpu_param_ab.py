"""GPU box: env-steps/s of the stepper with one blob parameter changed (A/B of model parameters, same process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.vec_env import FeedingJacoVecEnv
base = ModelBlob.load()
n, K = 4096, 300
for name, val in [(None, None)] + [(a.split('=')[0], float(a.split('=')[1])) for a in sys.argv[1:]] + [(None, None)]:
    blob = base if name is None else base.set_param(name, val)
    env = FeedingJacoVecEnv(n, pool_size=128, seed=1001, blob=blob)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    tape = torch.rand((K + 20, n, 7), device='cuda', generator=g) * 2 - 1
    for k in range(20): env.step(tape[k])
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(20, 20 + K): env.step(tape[k])
    torch.cuda.synchronize(); dt = time.time() - t0
    print(name, val, 'env-steps/s %.0f' % (n * K / dt), 'mean reward %.4f' % float(env.reward.mean()))
    env.close()
