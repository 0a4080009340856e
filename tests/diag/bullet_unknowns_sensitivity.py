"""What the [BULLET-UNVERIFIED] solver conventions are worth, measured in the CPU oracle: the switches of
include/agx_blob.h -- the residual early-out of the sweeps (oracle only), a second friction direction, warm-started contact normals, the
persistent 4-point manifold, the split-impulse threshold (oracle and device since round 4).  For each task workload, random-policy episodes are run with the conventions the device implements; from
every pre-step state the SAME step is repeated with one switch on, and the deviations of what the step returns are recorded:
reward, total_force_on_human, the tool's force, the largest observation entry.  Free-running 200-step episodes per switch give the
difference of the episode return beside it (dominated by how chaotic the scene is: the food pile).

    python tests/diag/bullet_unknowns_sensitivity.py [episodes] -> profiles/r04/bullet_unknowns_sensitivity.json (+ a markdown table on stdout)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from assistive_gym_amd.blob import ModelBlob       # noqa: E402
from oracle_lib import Oracle                       # noqa: E402

SWITCHES = {'residual early-out 1e-7': dict(ORACLE_RESIDUAL_EPS=1e-7), 'two friction directions': dict(FRICTION_DIRS=2),
            'warm start 0.85': dict(WARMSTART=0.85), 'persistent manifold': dict(MANIFOLD=1.0), 'split impulse below 4 cm': dict(SPLIT_PEN=0.04),
            'all five': dict(ORACLE_RESIDUAL_EPS=1e-7, FRICTION_DIRS=2, WARMSTART=0.85, MANIFOLD=1.0, SPLIT_PEN=0.04)}


def variant(blob, **kw):
    b = blob
    for k, v in kw.items():
        b = b.set_param(k, v)
    return b


def workloads(n):
    import bench
    from assistive_gym_amd.host.reset import make_states as feeding_states
    fj = ModelBlob.load('feeding_jaco')
    st, _ = feeding_states(fj, n, seed=4101)
    o = Oracle(fj)
    for s in st:
        o.settle(s, 25)
    yield 'FeedingJaco-v1 (food on the spoon, random policy)', fj, st, 1.0
    bb = ModelBlob.load('bed_bathing_sawyer')
    yield 'BedBathingSawyer-v1 (pad pressed on the arm, small actions)', bb, bench.wiping_pool(bb, n, 4201), 0.25
    from test_scratch_itch import scratching_state
    si = ModelBlob.load('scratch_itch_pr2').coop()
    osi = Oracle(si)
    yield 'ScratchItchPR2Human-v1 (tip pressed on the target, small actions)', si, np.stack([scratching_state(si, osi, seed=4301 + k) for k in range(n)]), 0.25


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = 200
    out = {'episodes_per_workload': n, 'steps': steps, 'workloads': {}}
    for name, blob, states, ascale in workloads(n):
        base = Oracle(blob)
        warm_clear = base.L.agxo_warm_clear
        tog = {k: Oracle(variant(blob, **kw)) for k, kw in SWITCHES.items()}
        dev = {k: dict(reward=[], force=[], tool=[], obs=[]) for k in SWITCHES}
        ret = {k: [] for k in list(SWITCHES) + ['device conventions']}
        rng = np.random.RandomState(7)
        acts = rng.uniform(-1, 1, (n, steps, blob.act_dim)).astype(np.float32) * ascale
        contacts = []
        for e in range(n):
            s = states[e].copy(); total = 0.0
            for t in range(steps):
                pre = s.copy()
                obs, rew, done, info = base.step(s, acts[e, t])
                total += rew; contacts.append(info[6])
                for k, o in tog.items():
                    if warm_clear:
                        warm_clear()
                    s2 = pre.copy()
                    o2, r2, d2, i2 = o.step(s2, acts[e, t])
                    dev[k]['reward'].append(abs(r2 - rew)); dev[k]['force'].append(abs(i2[0] - info[0])); dev[k]['tool'].append(abs(i2[3] - info[3]))
                    dev[k]['obs'].append(float(np.abs(np.asarray(o2) - np.asarray(obs)).max()))
                if done:
                    break
            ret['device conventions'].append(total)
            for k, o in tog.items():                                   # free-running episode with the switch on
                if warm_clear:
                    warm_clear()
                s2 = states[e].copy(); tot = 0.0
                for t in range(steps):
                    _, r2, d2, _ = o.step(s2, acts[e, t]); tot += r2
                    if d2:
                        break
                ret[k].append(tot)
        w = {'contacts_per_substep_mean': float(np.mean(contacts)), 'episode_return_device_conventions': ret['device conventions'], 'switches': {}}
        for k in SWITCHES:
            d = dev[k]
            w['switches'][k] = {q: dict(median=float(np.median(d[q])), p99=float(np.percentile(d[q], 99)), max=float(np.max(d[q]))) for q in d}
            w['switches'][k]['episode_return'] = ret[k]
            w['switches'][k]['episode_return_diff_mean'] = float(np.mean(np.array(ret[k]) - np.array(ret['device conventions'])))
        out['workloads'][name] = w
        print('\n### %s  (%.2f contacts per substep; episode return %s)' % (name, w['contacts_per_substep_mean'], np.round(ret['device conventions'], 2)))
        print('| switch | reward: median / p99 / max per step | total_force_on_human p99 / max (N) | tool force p99 / max (N) | obs max | episode return diff (mean of %d) |' % n)
        print('|---|---|---|---|---|---|')
        for k in SWITCHES:
            x = w['switches'][k]
            print('| %s | %.2e / %.2e / %.2e | %.2e / %.2e | %.2e / %.2e | %.2e | %+.3f |' % (k, x['reward']['median'], x['reward']['p99'], x['reward']['max'], x['force']['p99'], x['force']['max'],
                                                                                     x['tool']['p99'], x['tool']['max'], x['obs']['max'], x['episode_return_diff_mean']))
    os.makedirs(os.path.join(ROOT, 'profiles', 'r03'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r04', 'bullet_unknowns_sensitivity.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
