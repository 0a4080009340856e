"""Diagnostic (GPU box): first-substep contacts of the HIP stepper vs the oracle for one saved state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper, load
from oracle_lib import Oracle
np.set_printoptions(precision=7, suppress=True, linewidth=220)
blob = ModelBlob.load(); o = Oracle(blob)
s = np.load(os.path.join(ROOT, 'tools', 'dev_state_1_17.npy')); a = np.load(os.path.join(ROOT, 'tools', 'dev_action_1_17.npy'))
n = 1
st = Stepper(blob, n); st.set_state(s[None])
dev = torch.device('cuda', 0)
act = torch.from_numpy(a[None]).to(dev); obs = torch.zeros((n, 25), device=dev); rew = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, 8), device=dev)
dw = load().agx_debug_words(); dbg = torch.zeros((n, dw), device=dev)
st.step_dev(act, obs, rew, done, info, debug=dbg); torch.cuda.synchronize()
dbg = dbg.cpu().numpy()[0]
nc = int(dbg[0]); ce = dbg[16:16 + 1024].reshape(64, 16)[:nc]; cei = ce.view(np.int32)
con = o.substep_debug(s.copy())
print('gpu ncon', nc, 'oracle', len(con))
for k in range(nc):
    c = con[k] if k < len(con) else None
    bad = c is None or int(c[0]) != cei[k, 0] or int(c[1]) != cei[k, 1] or abs(c[11] - ce[k, 13]) > 1e-5 or np.abs(np.array(c[2:5]) - ce[k, 4:7]).max() > 1e-4
    if bad or cei[k, 0] < 13:
        print('contact', k, 'gpu', cei[k, 0], cei[k, 1], 'pa', ce[k, 4:7], 'n', ce[k, 10:13], 'dist', ce[k, 13])
        if c is not None: print('        oracle', int(c[0]), int(c[1]), 'pa', np.array(c[2:5]), 'n', np.array(c[8:11]), 'dist', c[11])
got = st.get_state()[0]
s1 = s.copy(); o.step(s1, a)
print('dq', np.abs(blob.view(got)['q'] - blob.view(s1)['q'])[0])

# per-substep comparison with FRAME_SKIP = 1 (targets are then re-derived each substep on both sides alike)
b1 = blob.set_param('FRAME_SKIP', 1)
o1 = Oracle(b1); st1 = Stepper(b1, 1)
cur = s.copy()
for k in range(5):
    st1.set_state(cur[None])
    dbg.zero_() if hasattr(dbg, 'zero_') else None
    dbg_t = torch.zeros((1, dw), device=dev)
    st1.step_dev(act, obs, rew, done, info, debug=dbg_t); torch.cuda.synchronize()
    g = st1.get_state()[0]
    d = dbg_t.cpu().numpy()[0]
    nxt = cur.copy(); r = o1.step(nxt, a)
    con = o.substep_debug(cur.copy())
    print('substep', k, 'gpu ncon', int(d[0]), 'oracle ncon(zero-action)', len(con), 'dq', np.abs(blob.view(g)['q'] - blob.view(nxt)['q'])[0, :8])
    ncg = int(d[0]); ce = d[16:16 + 1024].reshape(64, 16)[:ncg]; cei = ce.view(np.int32)
    for q in range(ncg):
        if cei[q, 0] < 13: print('    gpu robot contact', cei[q, 0], cei[q, 1], 'pa', ce[q, 4:7], 'n', ce[q, 10:13], 'dist', ce[q, 13], 'lam', ce[q, 14])
    for c in con:
        if int(c[0]) < 13: print('    orc robot contact', int(c[0]), int(c[1]), 'pa', np.array(c[2:5]), 'n', np.array(c[8:11]), 'dist', c[11], 'lam', c[12])
    cur = nxt

print('--- full step with FRAME_SKIP = k')
for k in range(1, 6):
    bk = blob.set_param('FRAME_SKIP', k)
    ok = Oracle(bk); stk = Stepper(bk, 1)
    stk.set_state(s[None])
    stk.step_dev(act, obs, rew, done, info); torch.cuda.synchronize()
    g = stk.get_state()[0]
    nxt = s.copy(); ok.step(nxt, a)
    print('k', k, 'dq', np.abs(blob.view(g)['q'] - blob.view(nxt)['q'])[0, :8], 'ncon', info.cpu().numpy()[0, 6])

print('--- per-substep with the exact targets of the 5-substep step (settle(1) from the oracle trajectory)')
full = s.copy(); o.step(full, a)
cur = s.copy(); blob.view(cur)['qt'][:] = blob.view(full)['qt']
st5 = Stepper(blob, 1)
for k in range(5):
    st5.set_state(cur[None]); st5.settle(1); st5.synchronize()
    g = st5.get_state()[0]
    nxt = cur.copy(); o.settle(nxt, 1)
    print('substep', k, 'dq', np.abs(blob.view(g)['q'] - blob.view(nxt)['q'])[0, :8], 'dqd', np.abs(blob.view(g)['qd'] - blob.view(nxt)['qd'])[0, :8].max())
    cur = nxt
print('oracle trajectory end vs oracle.step', np.abs(blob.view(cur)['q'] - blob.view(full)['q']).max())

print('--- GPU full step under tiny input perturbations (deviation from the oracle run on the same perturbed input)')
rng = np.random.RandomState(3)
stp = Stepper(blob, 1)
for t in range(12):
    sp = s.copy(); v = blob.view(sp)
    if t > 0:
        v['q'][0, :10] += rng.uniform(-2e-6, 2e-6, 10).astype(np.float32)
        v['qd'][0, :10] += rng.uniform(-2e-6, 2e-6, 10).astype(np.float32)
    stp.set_state(sp[None]); stp.step_dev(act, obs, rew, done, info); torch.cuda.synchronize()
    g = stp.get_state()[0]
    so = sp.copy(); o.step(so, a)
    print(t, 'dq max', np.abs(blob.view(g)['q'] - blob.view(so)['q']).max(), 'ncon gpu', info.cpu().numpy()[0, 6])
