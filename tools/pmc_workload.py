"""Workload for the PMC passes: 6 observation-only launches (known traffic: one state record read,
one observation written per env) followed by 6 full env.step launches, 4096 envs, unchunked
(AGX_CHUNKS=1: every launch covers all environments, so per-launch counters are per 4096 environments).
  python tools/pmc_workload.py [feeding|bedbathing|scratchitch|armmanipulation|dressing|drinking] [envs]
(dressing: 1024 environments by default -- one 1,024-thread cloth workgroup each -- and 4 steps)"""
import os, sys
os.environ.setdefault('AGX_CHUNKS', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from assistive_gym_amd import vec_env
task = sys.argv[1] if len(sys.argv) > 1 else 'feeding'
cls = {'feeding': 'FeedingJacoVecEnv', 'bedbathing': 'BedBathingSawyerVecEnv', 'scratchitch': 'ScratchItchPR2HumanVecEnv', 'armmanipulation': 'ArmManipulationSawyerVecEnv',
       'dressing': 'DressingBaxterVecEnv', 'drinking': 'DrinkingJacoVecEnv'}[task]
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if task == 'dressing' else 4096)
env = getattr(vec_env, cls)(n, pool_size=16 if task == 'dressing' else 32 if task in ('scratchitch', 'armmanipulation') else 64, seed=1001, auto_reset=False)
env.reset()
torch.cuda.synchronize()
for _ in range(6):
    env.stepper.observe_dev(env.obs)
torch.cuda.synchronize()
g = torch.Generator(device='cuda'); g.manual_seed(1)
for _ in range(4 if task == 'dressing' else 6):
    a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
    env.step(a)
torch.cuda.synchronize()
