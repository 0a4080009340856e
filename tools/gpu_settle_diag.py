"""GPU diagnostic: single settle steps of the bed_settle model on the device against the oracle, with the contact sets of any
environment that disagrees.  python tools/gpu_settle_diag.py [friction] -> gpurun_out/settle_diag.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.model import compiler as L
from oracle_lib import Oracle
from test_bed_settle import posed

mu = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sb0, bb = ModelBlob.load('bed_settle'), ModelBlob.load('bed_bathing_sawyer')
w = sb0.words.copy()
for c in range(*sb0.meta['ranges']['bed']):
    w.view(np.float32)[sb0.h['OFF_COLL'] + c * L.C['STRIDE'] + L.C['FRICTION']] = mu
sb = ModelBlob(w, sb0.meta)
o = Oracle(sb)
N = 8
st = Stepper(sb, N)
lay = st.debug_layout()
DW, DCON = lay[0], lay[1]
dbg = torch.zeros(N, DW, device='cuda')
ref = np.array([posed(sb, bb, 7001 + i)[0] for i in range(N)])
np.set_printoptions(precision=6, suppress=True, linewidth=200)
out = open(os.path.join(ROOT, 'gpurun_out', 'settle_diag.txt'), 'w')
total = 0
for adv in [0] + [5] * 24:
    for i in range(N):
        o.settle(ref[i], adv)
    total += adv
    got2 = []
    for rep in range(2):
        st.set_state(ref)
        st.settle_debug(1, dbg)
        torch.cuda.synchronize()
        got2.append(st.get_state())
    got = got2[0]
    nxt = ref.copy()
    cons = []
    for i in range(N):
        cons.append(o.substep_debug(nxt[i]))
    d = np.abs(got[:, :47] - nxt[:, :47]).max(1)
    print('step', total, 'dq max per env', d, 'repeatable', np.array_equal(got2[0], got2[1]), file=out)
    D = dbg.cpu().numpy()
    for i in range(N):
        if d[i] > 5e-5:
            nc = int(D[i, 0]); ce = D[i, DCON:DCON + 1024].reshape(64, 16)[:nc]
            pe = [(int(x), int(y)) for x, y in ce.view(np.int32)[:, :2]]; po = [(int(c[0]), int(c[1])) for c in cons[i]]
            print('  env', i, 'device nc/rows/overflow', D[i, :3], 'oracle nc', len(po), file=out)
            print('  device-only', sorted(set(pe) - set(po)), 'oracle-only', sorted(set(po) - set(pe)), file=out)
            for k, pr in enumerate(po):
                if pr in pe:
                    j = pe.index(pr)
                    print('   ', pr, 'dist dev/oracle', ce[j, 13], cons[i][k][11], 'n', ce[j, 10:13], cons[i][k][8:11], file=out)
            np.save(os.path.join(ROOT, 'gpurun_out', 'settle_bad_state_%d_%d.npy' % (total, i)), ref[i])
            np.save(os.path.join(ROOT, 'gpurun_out', 'settle_bad_got_%d_%d.npy' % (total, i)), got[i])
out.close()
print(open(os.path.join(ROOT, 'gpurun_out', 'settle_diag.txt')).read()[-6000:])
