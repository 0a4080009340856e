"""GPU box: FeedingJaco at 4096 environments under the random policy of tests/test_gpu_parity.py::test_full_episode_invariants_at_bench_size; every
environment that ends before step 200 (the non-finite guard) is saved with the state it started the step from and its action, for replay on the
CPU wave emulator.  usage: python tools/gpu_early_done_cases.py [VecEnvClass] [max_cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd import vec_env
cls = sys.argv[1] if len(sys.argv) > 1 else 'FeedingJacoVecEnv'
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 4096
env = getattr(vec_env, cls)(n, pool_size=64, seed=1001)
env.reset(); env.auto_reset = False
g = torch.Generator(device='cuda'); g.manual_seed(11)
st = env.stepper.state_tensor()
cases, total = [], 0
hist = []
for k in range(199):
    prev = st.clone()
    a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
    obs, rew, done, info = env.step(a)
    hist.append((prev, a))
    hist = hist[-3:]
    d = done.bool()
    if bool(d.any()):
        idx = torch.nonzero(d).flatten().cpu().numpy()
        total += len(idx)
        for i in idx:
            if len(cases) < cap:
                cases.append(dict(env=int(i), step=k, states=np.stack([h[0][i].cpu().numpy() for h in hist]), actions=np.stack([h[1][i].cpu().numpy() for h in hist]), info=info[i].cpu().numpy()))
        # the guard ended them; put them back on a pool state so that they are not counted again
        env.stepper.reset_done(env.pool, env.pool_size, env.done, 0, iteration=k + 1)
print('%s: %d early-done events in %d env steps; pool entries of the first cases: %s' % (cls, total, n * 199, [c['env'] % 64 for c in cases]))
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r05d'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'r05d', 'early_done_cases_%s.npz' % cls), envs=np.array([c['env'] for c in cases]), steps=np.array([c['step'] for c in cases]),
         states=np.array([c['states'] for c in cases if len(c['states']) == 3] or [np.zeros(0)]), actions=np.array([c['actions'] for c in cases if len(c['actions']) == 3] or [np.zeros(0)]))
