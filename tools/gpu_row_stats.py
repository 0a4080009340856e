"""Distribution of solver rows / contacts per substep-end over a random-policy episode (sizes the row-space PGS storage)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd import vec_env
out = {}
for task, cls in (('feeding', 'FeedingJacoVecEnv'), ('bedbathing', 'BedBathingSawyerVecEnv')):
    n = 4096
    env = getattr(vec_env, cls)(n, pool_size=256, seed=1001)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    rows, cons = [], []
    for k in range(400):
        a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
        obs, rew, done, info = env.step(a)
        if k % 5 == 0:
            rows.append(info[:, 7].cpu().numpy().copy()); cons.append(info[:, 6].cpu().numpy().copy())
    rows, cons = np.concatenate(rows), np.concatenate(cons)
    q = [50, 90, 99, 99.9, 100]
    out[task] = dict(rows_mean=float(rows.mean()), rows_pct=dict(zip(map(str, q), np.percentile(rows, q).tolist())), contacts_mean=float(cons.mean()),
                     contacts_pct=dict(zip(map(str, q), np.percentile(cons, q).tolist())), overflow=env.stepper.overflow_count(),
                     frac_rows_gt_128=float((rows > 128).mean()), frac_rows_gt_136=float((rows > 136).mean()))
    env.close()
print(json.dumps(out, indent=1))
