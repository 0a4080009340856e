"""GPU box: how many rows does a step of the wide row-local sweep (csrc/agx_pgs_lvw.h) visit?  Debug launches (agx_step_debug) of FeedingJaco
environments along a random-policy rollout; the solve kernel of the first substep leaves in the debug record: steps executed, rows visited,
steps of the two static schedules, rows of the substep.  Prints one JSON line.
usage: python tools/gpu_solve_streams.py [n_envs] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd import vec_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
env = vec_env.FeedingJacoVecEnv(n, pool_size=64, seed=1001)
env.reset()
lay = env.stepper.debug_layout()
T = lay[6]
g = torch.Generator(device='cuda'); g.manual_seed(1)
dbg = torch.zeros((n, lay[0]), device='cuda')
acc = np.zeros(4); cnt = 0
for k in range(steps):
    a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
    if k % 5 == 4:
        env.stepper.step_dev(a, env.obs, env.reward, env.done, env.info, torch.cuda.current_stream().cuda_stream, debug=dbg)
        env.stepper.reset_done(env.pool, env.pool_size, env.done, torch.cuda.current_stream().cuda_stream)
        d = dbg[:, T + 16:T + 20].cpu().numpy()
        ok = d[:, 0] > 0                      # environments that took the wide sweep in this substep
        acc += d[ok].sum(axis=0); cnt += int(ok.sum())
    else:
        env.step(a)
sweeps = int(env.blob.param('NITER'))
print(json.dumps(dict(n_envs=n, steps=steps, sampled_substeps=cnt, rows_per_substep=acc[3] / cnt, rows_visited_per_sweep=acc[1] / cnt / sweeps,
                      steps_executed_per_sweep=acc[0] / cnt / sweeps, rows_per_step=acc[1] / acc[0], static_steps_per_sweep=acc[2] / cnt,
                      visits_of_the_narrow_sweep_over_steps_of_the_wide=acc[1] / acc[0])))
