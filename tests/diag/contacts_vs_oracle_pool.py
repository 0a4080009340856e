import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper, load
from assistive_gym_amd.host.reset import make_states
from oracle_lib import Oracle
np.set_printoptions(precision=6, suppress=True, linewidth=200)
blob = ModelBlob.load(); o = Oracle(blob)
n = 4
states, _ = make_states(blob, n, seed=4001)
st = Stepper(blob, n); st.set_state(states)
dev = torch.device('cuda', 0)
act = torch.zeros((n, 7), device=dev); obs = torch.zeros((n, 25), device=dev); rew = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, 8), device=dev)
dw = load().agx_debug_words(); dbg = torch.zeros((n, dw), device=dev)
st.step_dev(act, obs, rew, done, info, debug=dbg); torch.cuda.synchronize()
dbg = dbg.cpu().numpy()
for i in range(2):
    nc = int(dbg[i, 0]); ce = dbg[i, 16:16 + 1024].reshape(64, 16)[:nc]; cei = ce.view(np.int32)
    con = o.substep_debug(states[i].copy())
    print('env', i, 'gpu ncon', nc, 'oracle', len(con), 'overflow', dbg[i, 2])
    print(' gpu   :', [(int(cei[k, 0]), int(cei[k, 1]), round(float(ce[k, 13]), 5)) for k in range(nc)])
    print(' oracle:', [(int(c[0]), int(c[1]), round(float(c[11]), 5)) for c in con])
    for k in range(nc):
        if cei[k, 1] == 208 or cei[k, 0] >= 77 and cei[k, 0] < 147:
            print('  gpu bowl/table contact', cei[k, 0], cei[k, 1], 'pa', ce[k, 4:7], 'pb', ce[k, 7:10], 'n', ce[k, 10:13], 'dist', ce[k, 13])
    v = blob.view(states[i]); print(' bowl state', v['free'][0, 1])
