"""GPU diagnostic: the packed solve kernel (agx_pgs4.h; an opt-in build: python -m assistive_gym_amd.build --extra "-DAGX_USE_SOLVE4=1" --out
assistive_gym_amd/lib/libagx_packed.so, then AGX_LIB=<that file>) against the one-wave-per-environment kernel (AGX_SOLVE=old) on the same states.
  python tools/gpu_p4_diag.py [model] [n] [steps]      -> prints deviations, saves offending input states under gpurun_out/p4_diag/"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.vec_env import build_reset_pool
model = sys.argv[1] if len(sys.argv) > 1 else 'feeding_panda'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
b = ModelBlob.load(model)
states = build_reset_pool(b, n, 5001)
os.environ['AGX_SOLVE'] = 'new'; sp = Stepper(b, n)
os.environ['AGX_SOLVE'] = 'old'; so = Stepper(b, n)
out = os.path.join(ROOT, 'gpurun_out', 'p4_diag'); os.makedirs(out, exist_ok=True)
rng = np.random.RandomState(7)
ref = states.copy()
for k in range(steps):
    a = rng.uniform(-1, 1, (n, b.act_dim)).astype(np.float32)
    sp.set_state(ref); so.set_state(ref)
    op, rp, dp, ip = sp.step_host(a); oo, ro, do, io = so.step_host(a)
    gp, go = sp.get_state(), so.get_state()
    for i in range(n):
        vp, vo = b.view(gp[i]), b.view(go[i])
        dq = np.abs(vp['q'] - vo['q']).max(); df = np.abs(vp['free'][0][:, :3] - vo['free'][0][:, :3]).max()
        fin = np.isfinite(gp[i][:b.h['S_BASE']]).all()
        if bool(dp[i]) != bool(do[i]) or not fin or dq > 1e-4 or df > 1e-3 or ip[i, 6] != io[i, 6]:
            print('step %d env %d: done %s/%s finite %s dq %.2e dfree %.2e info6 %s/%s nrows %s/%s' % (k, i, dp[i], do[i], fin, dq, df, ip[i, 6], io[i, 6], ip[i, 7], io[i, 7]))
            np.save(os.path.join(out, '%s_step%d_env%d_state.npy' % (model, k, i)), ref[i]); np.save(os.path.join(out, '%s_step%d_env%d_action.npy' % (model, k, i)), a[i])
            bad = np.nonzero(~np.isfinite(gp[i][:b.h['S_BASE']]))[0]
            if len(bad): print('   non-finite words', bad[:20])
    ref = go
print('diag done')
