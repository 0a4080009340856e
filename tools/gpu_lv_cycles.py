"""GPU box: shader-clock cycles of the Gauss-Seidel solve of the first substep per FeedingJaco environment (debug record), per visit of a row,
with the batch size as the knob: 256 environments = one wavefront per CU (the dependent chain of ONE wave), 4096 = sixteen per CU (what the
product runs: chain + contention).  usage: python tools/gpu_lv_cycles.py [n_envs ...]   (AGX_LIB / AGX_SOLVE_LDS_BYTES select the build / the window)"""
import os, sys
os.environ.setdefault('AGX_CHUNKS', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from assistive_gym_amd import vec_env
for n in [int(a) for a in sys.argv[1:]] or [256, 4096]:
    env = vec_env.FeedingJacoVecEnv(n, pool_size=64, seed=1001)
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
    rows, cyc, tail, nent = D[:, 1], D[:, lay[6] + 5], D[:, lay[6] + 6], D[:, 4]
    print('%s lds=%s n=%d: rows median %.0f, solve cycles median %.0f (p10 %.0f, p90 %.0f), per row and sweep %.0f; tail (integration, store) median %.0f'
          % (os.path.basename(os.environ.get('AGX_LIB', 'libagx.so')), os.environ.get('AGX_SOLVE_LDS_BYTES', 'default'), n, np.median(rows), np.median(cyc), np.percentile(cyc, 10),
             np.percentile(cyc, 90), np.median(cyc / (50 * np.maximum(rows, 1))), np.median(tail)))
    print('   pairs per environment: median %.0f, p75 %.0f, p90 %.0f, p99 %.0f, max %.0f' % (np.median(nent), np.percentile(nent, 75), np.percentile(nent, 90), np.percentile(nent, 99), nent.max()))
    env.close()
