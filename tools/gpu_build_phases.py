"""GPU box: shader-clock cycles of the build kernel's phases for the first substep of one step, per environment (debug record; every
environment is stepped with the debug kernels, i.e. at the debug path's occupancy): kinematics, ABA + M^-1, velocity prediction, collision
(with its broadphase / narrowphase / selection split), rows.   python tools/gpu_build_phases.py [VecEnv class]"""
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
T = dbg.cpu().numpy()[:, lay[6]:lay[6] + 16]
names = {0: 'kinematics', 1: 'ABA + M^-1', 2: 'predict velocities', 3: 'collide (all)', 4: 'rows', 5: 'PGS', 6: 'integrate + hooks + store', 7: 'row-space set-up (inside PGS)',
         8: 'collide: AABBs', 9: 'collide: group cull', 10: 'collide: sweep', 11: 'collide: narrowphase (GJK)', 12: 'collide: selection', 13: 'narrowphase pairs', 14: 'narrowphase passes'}
print(cls, 'contacts %.2f rows %.1f' % (dbg[:, 0].mean().item(), dbg[:, 1].mean().item()))
for k in sorted(names):
    print('%-32s median %9.0f  mean %9.0f  max %9.0f' % (names[k], np.median(T[:, k]), T[:, k].mean(), T[:, k].max()))
