"""Randomised emulator-vs-oracle campaign over EVERY model blob (CPU only, test infrastructure): a host-sampled post-reset state, a
random-policy rollout of random length on the oracle (co-op flavours and velocity kicks included), then one env.step() through the kernel
sources on the wave emulator and through the oracle; prints every case whose observation / reward / joint angles / contact counts differ.

    for w in $(seq 0 7); do python tests/diag/fuzz_models.py $w 30 & done; wait
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from emu_lib import Emu                        # noqa: E402
from oracle_lib import Oracle                  # noqa: E402
from assistive_gym_amd.blob import ModelBlob   # noqa: E402
from assistive_gym_amd.model import compiler as L   # noqa: E402

w, N = int(sys.argv[1]), int(sys.argv[2])
names = sys.argv[3:] or sorted(f[:-8] for f in os.listdir(os.path.join(ROOT, 'assistive_gym_amd', 'data')) if f.endswith('.agxblob') and f != 'bed_settle.agxblob')
rng = np.random.RandomState(7000 + w)
cache = {}


def states(b, seed):
    k = b.task_kind
    if k == L.TASK_FEEDING:
        from assistive_gym_amd.host.reset import make_states
        return make_states(b, 1, seed=seed)[0]
    if k == L.TASK_BED_BATHING:
        from assistive_gym_amd.host.reset_bed import make_states
        return make_states(b, 1, seed=seed)[0]
    if k == L.TASK_SCRATCH_ITCH:
        from assistive_gym_amd.host.reset_scratch import make_states
        return make_states(b, 1, seed=seed)[0]
    if k == L.TASK_ARM_MANIPULATION:
        from assistive_gym_amd.host.reset_arm import make_states
        return make_states(b, 1, seed=seed)[0]
    from assistive_gym_amd.host.reset_dressing import make_states       # the rigid scene of the dressing models (no garment attached)
    return make_states(b, 1, seed=seed)[0]


bad = 0
for t in range(N):
    name = names[rng.randint(len(names))]
    coop = rng.rand() < 0.3
    key = (name, coop)
    if key not in cache:
        b = ModelBlob.load(name)
        b = b.coop() if coop else b
        cache[key] = (b, Oracle(b), Emu(b))
    b, o, e = cache[key]
    seed = int(rng.randint(1, 1 << 30))
    st = states(b, seed)[0].copy()
    if b.task_kind == L.TASK_FEEDING:
        o.settle(st, 25)
    nsteps = int(rng.randint(0, 25 if b.task_kind != L.TASK_DRESSING else 4))
    scale = rng.choice([1.0, 1.0, 3.0])
    for k in range(nsteps):
        o.step(st, (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32))
    v = b.view(st[None])
    if rng.rand() < 0.3:
        v['qd'][0, :b.nrobot] += rng.uniform(-1, 1, b.nrobot)
    if not np.isfinite(st[:b.h['S_ENV']]).all():
        print('NONFINITE after the oracle rollout', name, seed, flush=True)
        continue
    a = (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32)
    so, se = st.copy(), st.copy()
    oo, eo = o.step(so, a), e.step(se, a)
    vo, ve = b.view(so[None]), b.view(se[None])
    dev = dict(obs=float(np.abs(oo[0] - eo[0]).max()), rew=float(abs(oo[1] - eo[1])), q=float(np.abs(vo['q'] - ve['q']).max()),
               ncon=float(abs(oo[3][6] - eo[3][6])), nrow=float(abs(oo[3][7] - eo[3][7])), force=float(abs(oo[3][0] - eo[3][0]) / max(1.0, abs(oo[3][0]))))
    force_cols = 3 if b.task_kind in (L.TASK_ARM_MANIPULATION,) else 2
    obs_dev = np.abs(oo[0] - eo[0]); obs_dev[-force_cols - (0 if not coop else 0):] = 0
    flag = dev['q'] > 5e-5 or dev['ncon'] > 0 or dev['force'] > 2e-3 or (dev['ncon'] == 0 and oo[3][6] == 0 and dev['obs'] > 1e-4)
    if flag:
        bad += 1
        print('MISMATCH worker', w, 'case', t, name, 'coop', coop, 'seed', seed, 'nsteps', nsteps, 'scale', scale, 'ncon', oo[3][6], dev, flush=True)
        np.save('/tmp/fuzzm_bad_%d_%d.npy' % (w, t), np.concatenate([st, a]))
print('worker', w, 'done', N, 'cases, mismatches', bad, flush=True)
