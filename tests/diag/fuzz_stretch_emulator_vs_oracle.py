"""Randomised emulator-vs-oracle campaign for the Stretch (CPU only, test infrastructure): states sampled by the reset generator's numpy twin,
settled, advanced by random-policy rollouts of random length in the oracle (co-op and impairments included, actions up to 3x the box), then ONE
env.step() through the kernel sources on the wave emulator and through the oracle; prints the cases whose observation / reward / joint angles /
contact counts disagree and a summary line.

    for w in $(seq 0 7); do python tests/diag/fuzz_stretch_emulator_vs_oracle.py $w 6 & done; wait"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from emu_lib import Emu                        # noqa: E402
from oracle_lib import Oracle                  # noqa: E402
import reset_oracle as ro                      # noqa: E402
from assistive_gym_amd.blob import ModelBlob   # noqa: E402

w, N = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(7000 + w)
bad, worst = 0, dict(obs=0.0, q=0.0, rew=0.0)
for t in range(N):
    model = ['feeding_stretch', 'scratch_itch_stretch'][int(rng.randint(2))]
    base = ModelBlob.load(model)
    b = base.coop() if rng.rand() < 0.3 else base
    o, e = Oracle(b), Emu(b)
    seed = int(rng.randint(1, 1 << 30)); imp = int(rng.choice([-1, 3, 1]))
    st, _ = ro.ResetOracle(b.words).sample(seed, imp)
    o.settle(st, 25 if model.startswith('feeding') else 10)
    scale = rng.choice([1.0, 1.0, 3.0])
    for k in range(int(rng.randint(0, 50))):
        o.step(st, (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32))
    if not np.isfinite(st[:b.h['S_ENV']]).all():
        print('worker', w, 'case', t, model, 'oracle rollout went non-finite (seed %d)' % seed); bad += 1; continue
    a = (rng.uniform(-1, 1, b.act_dim) * scale).astype(np.float32)
    s1, s2 = st.copy(), st.copy()
    oo = o.step(s1, a); ee = e.step(s2, a)
    dq = float(np.abs(b.view(s1[None])['q'] - b.view(s2[None])['q']).max())
    dobs = float(np.abs(oo[0] - ee[0]).max()); drew = abs(float(oo[1]) - float(ee[1]))
    worst = dict(obs=max(worst['obs'], dobs), q=max(worst['q'], dq), rew=max(worst['rew'], drew))
    if dobs > 3e-4 or dq > 3e-4 or drew > 3e-4 or oo[3][6] != ee[3][6]:
        bad += 1
        print('worker', w, 'case', t, model, 'coop' if b.is_coop else 'solo', 'seed', seed, 'obs %.2e q %.2e rew %.2e contacts %d / %d' % (dobs, dq, drew, oo[3][6], ee[3][6]), flush=True)
print('worker', w, 'cases', N, 'disagreeing', bad, 'worst', {k: '%.1e' % v for k, v in worst.items()}, flush=True)
