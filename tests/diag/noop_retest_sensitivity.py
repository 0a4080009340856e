"""What does the no-op re-test rule of the PGS (AGX_P_NOOP_RETEST = K, oracle pgs() / csrc/agx_pgs.h) change?  CPU oracle, FeedingJaco,
random-policy episodes: every env.step of the K = 0 trajectory (plain PGS) is repeated from the same state with K = 2, 5, 10, and -- for
scale -- with 49 / 60 / 100 sweeps instead of 50.  Writes profiles/r03/noop_retest_sensitivity.json.

    python tests/diag/noop_retest_sensitivity.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from assistive_gym_amd.blob import ModelBlob            # noqa: E402
from assistive_gym_amd.host.reset import make_states    # noqa: E402
from oracle_lib import Oracle                           # noqa: E402

b0 = ModelBlob.load('feeding_jaco').set_param('NOOP_RETEST', 0)
o0 = Oracle(b0)
variants = {'K=2': b0.set_param('NOOP_RETEST', 2), 'K=5': b0.set_param('NOOP_RETEST', 5), 'K=10': b0.set_param('NOOP_RETEST', 10),
            'sweeps=49': b0.set_param('NITER', 49), 'sweeps=60': b0.set_param('NITER', 60), 'sweeps=100': b0.set_param('NITER', 100)}
orc = {k: Oracle(v) for k, v in variants.items()}
stats = (C.c_double * 8).in_dll(o0.L, 'g_pgs_stats')
st, _ = make_states(b0, 8, seed=277)
rng = np.random.RandomState(0)
dev = {k: dict(q=[], free=[], rew=[], force=[], flips=0, visits=0.0) for k in variants}
base_visits, n = 0.0, 0


def visits():
    v = list(stats)
    return (v[0] + v[2] - v[4]) / max(v[5], 1) / 50      # row visits per sweep; exact friction no-ops are skipped by the device in any case


for i in range(8):
    s = st[i].copy(); o0.settle(s, 25)
    for k in range(40):
        a = rng.uniform(-1, 1, 7).astype(np.float32)
        s_in = s.copy()
        for q in range(8): stats[q] = 0
        ob, r, d, info = o0.step(s, a)
        base_visits += visits(); n += 1
        v1 = b0.view(s.reshape(1, -1))
        near = np.abs(v1['free'][0][:, :3]).max(axis=1) < 100
        for name, o in orc.items():
            s2 = s_in.copy()
            for q in range(8): stats[q] = 0
            ob2, r2, d2, info2 = o.step(s2, a)
            e = dev[name]; e['visits'] += visits() * (50.0 / o.blob.param('NITER'))
            v2 = b0.view(s2.reshape(1, -1))
            e['q'].append(float(np.abs(v1['q'] - v2['q']).max())); e['free'].append(float(np.abs(v1['free'][0][near, :3] - v2['free'][0][near, :3]).max()))
            e['rew'].append(abs(r - r2)); e['force'].append(abs(float(info[0]) - float(info2[0])))
            e['flips'] += int(v1['food_alive'][0] != v2['food_alive'][0])
out = dict(workload='FeedingJaco-v1, 8 environments x 40 random-policy steps, CPU oracle (f64); single-step deviations from the plain 50-sweep PGS started from the same state',
           steps=n, visits_per_sweep_plain=base_visits / n, variants={})
for name, e in dev.items():
    out['variants'][name] = dict(visits_per_sweep=e['visits'] / n,
                                 joint_angle_dev=dict(max=max(e['q']), p99=float(np.percentile(e['q'], 99)), median=float(np.median(e['q']))),
                                 free_body_position_dev_m=dict(max=max(e['free']), p99=float(np.percentile(e['free'], 99)), median=float(np.median(e['free']))),
                                 reward_dev=dict(max=max(e['rew']), p99=float(np.percentile(e['rew'], 99))),
                                 total_force_dev=dict(max=max(e['force'])), food_event_flips=e['flips'])
    print(name, json.dumps(out['variants'][name]))
os.makedirs(os.path.join(ROOT, 'profiles', 'r03'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r03', 'noop_retest_sensitivity.json'), 'w'), indent=1)
