"""Oracle soak (CPU only, test infrastructure): a 200-step random-policy episode of every model without a garment, single-agent and co-op,
from a host-sampled state; prints return, peak force on the human and FAIL on a non-finite value.   python tests/diag/soak_oracle_all_models.py <worker 0..3>"""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import time
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.model import compiler as L
from oracle_lib import Oracle
names = sorted(f[:-8] for f in os.listdir(os.path.join(ROOT, 'assistive_gym_amd', 'data')) if f.endswith('.agxblob') and f != 'bed_settle.agxblob' and not f.startswith('dressing') and not f.startswith('drinking'))
w = int(sys.argv[1]); names = names[w::4]
def states(b, seed):
    k = b.task_kind
    if k == L.TASK_FEEDING:
        from assistive_gym_amd.host.reset import make_states; st = make_states(b, 1, seed=seed)[0]; Oracle(b).settle(st[0], 25); return st[0]
    if k == L.TASK_BED_BATHING:
        from assistive_gym_amd.host.reset_bed import make_states; return make_states(b, 1, seed=seed)[0][0]
    if k == L.TASK_SCRATCH_ITCH:
        from assistive_gym_amd.host.reset_scratch import make_states; return make_states(b, 1, seed=seed)[0][0]
    from assistive_gym_amd.host.reset_arm import make_states
    fo = Oracle(b.set_param('HUMAN_GRAVITY_Z', -1.0))
    def fall(st, n):
        st = st.copy(); [fo.settle(st[i], n) for i in range(len(st))]; return st
    return make_states(b, 1, seed=seed, arm_settler=fall)[0][0]
for name in names:
    for coop in (False, True):
        b = ModelBlob.load(name); b = b.coop() if coop else b
        o = Oracle(b)
        t = time.time(); ret = 0.0; maxf = 0.0; ok = True
        s = states(b, 4242).copy()
        rng = np.random.RandomState(1)
        for k in range(200):
            obs, rew, done, info = o.step(s, rng.uniform(-1, 1, b.act_dim).astype(np.float32))
            ret += rew; maxf = max(maxf, float(info[0]))
            if not (np.isfinite(obs).all() and np.isfinite(rew)): ok = False; break
        print(name, 'coop' if coop else 'solo', 'ok' if ok and done else 'FAIL at %d' % k, 'return %.1f' % ret, 'max force %.1f' % maxf, '%.1fs' % (time.time() - t), flush=True)
