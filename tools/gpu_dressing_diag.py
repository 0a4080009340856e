"""GPU diagnostic: the dressing model (rigid substeps + cloth kernel) on the device against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.host import reset_dressing as rd
from oracle_lib import Oracle

b = ModelBlob.load('dressing_baxter')
o = Oracle(b)
N = 4
st, cloth, infos = rd.make_states(b, N, seed=11)
v = b.view(st); v['task'][:, 0] = np.array([-9.81 / 2], dtype=np.float32).view(np.int32)[0]
dev = Stepper(b, N)
print('variant', dev.variant(), 'cloth nodes', dev.cloth_nodes())
dev.set_state(st); dev.set_cloth(cloth)
ref_s, ref_c = st.copy(), cloth.copy()
for k in range(6):
    t = time.time(); dev.settle(1); dev.L.agx_synchronize(dev.h, None); td = time.time() - t
    gs, gc = dev.get_state(), dev.get_cloth()
    for i in range(N):
        o.settle_cloth(ref_s[i], ref_c[i], 1)
    dx = np.abs(gc[:, 0] - ref_c[:, 0]); dv = np.abs(gc[:, 1] - ref_c[:, 1])
    print('sim step', k, 'device %.1f ms' % (td * 1e3), 'state diff', np.abs(gs - ref_s).max(), 'cloth x: max %.2e p99 %.2e median %.2e' % (dx.max(), np.percentile(dx, 99), np.median(dx)),
          'v: max %.2e p99 %.2e' % (dv.max(), np.percentile(dv, 99)), 'finite', np.isfinite(gc).all())
    # continue both from the oracle's state so that differences do not accumulate
    dev.set_state(ref_s); dev.set_cloth(ref_c)
