"""Why does BedBathingSawyer need the conditioning ladder?  

At the bench size, config 3 under the random policy had 8 of 64 compared environments whose pose deviated beyond 1e-4 (worst 2.35e-3) and were
judged against the oracle's own 1-ulp sensitivity: environments whose arm rests on the mattress or the person.  The stated cause -- one GJK
witness point per hull pair sliding over a resting contact, 50 cold sweeps on a stiff contact -- is a property of THIS solver; Bullet keeps a
persistent 4-point manifold and warm-starts its impulses.  Both exist here as blob switches (AGX_P_MANIFOLD, AGX_P_WARMSTART, oracle and
device).  This script measures, with nothing but the f64 oracle, the 1-ulp sensitivity of the pose block of the observation after ONE step
from states of a random-policy rollout, for the default conventions and with the switches on (the caches warmed by the rollout's last steps
and restored before every trial), and reports how many environments would need a conditioning level (K x sensitivity > 1e-4) either way.
usage: python tests/diag/resting_contact_sensitivity.py [--envs 48] [--steps 20] [--out profiles/r06/resting_contact_sensitivity.json]
       python tests/diag/resting_contact_sensitivity.py --from profiles/r06/bench_size_states_config3_random.npz [--out ...]
--from: the 64 environments the bench-size parity test compared on the GPU (tests/test_gpu_bench_size.py with AGX_DUMP_BENCH_STATES: states after the
4096-environment rollout, the next actions, the device's results): one step from each, cold caches, under the default conventions and with
the persistent manifold, warm starting, the friction direction fixed to the plane-space tangent (AGX_P_FRIC_EPS = 1e30: never the slip
direction) and two friction directions.  RESULT (round 6): the 8 ill-conditioned environments are limbs of the PERSON lying on the mattress
(human link x bed box: the face manifold is already there), not the robot; manifold and warm start change nothing; the velocity-dependent
friction direction is half of it -- with the fixed tangent 4 of the 8 fall to 1e-7, worst 2.4e-3 -> 7.7e-4."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from assistive_gym_amd.blob import ModelBlob            # noqa: E402
import conditioning as C                                 # noqa: E402
import oracle_lib                                        # noqa: E402


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def rollout_and_measure(blob, states, steps, trials, label):
    o = oracle_lib.Oracle(blob)
    f = blob.obs_dim_robot - 1
    res = []
    for i, s0 in enumerate(states):
        o.forget_warm()
        s = s0.copy()
        rng = np.random.RandomState(100 + i)
        for k in range(steps):
            o.step(s, rng.uniform(-1, 1, blob.act_dim).astype(np.float32))
        a = rng.uniform(-1, 1, blob.act_dim).astype(np.float32)
        man, warm = o.manifold_get(), o.warm_get()

        def run(st):
            o.manifold_set(man); o.warm_set(warm)
            st = st.copy()
            obs, rew, done, info = o.step(st, a)
            return np.asarray(obs, dtype=np.float64), float(rew), np.asarray(info, dtype=np.float64)
        C.IN_SENSITIVITY[0] = True
        o0, r0, i0 = run(s)
        fw = C.float_words(blob)
        prng = np.random.RandomState(i)
        pose, force, reward = 0.0, 0.0, 0.0
        for _ in range(trials):
            sp = s.copy(); sp[fw] = C._perturb_f32(sp[fw], prng)
            o1, r1, i1 = run(sp)
            d = np.abs(o1 - o0)
            pose = max(pose, float(np.delete(d, [f]).max())); force = max(force, float(abs(i1[0] - i0[0]))); reward = max(reward, abs(r1 - r0))
        C.IN_SENSITIVITY[0] = False
        res.append(dict(env=i, contacts=int(i0[6]) % 1000, pose_sens=pose, force_sens=force, reward_sens=reward, total_force=float(i0[0])))
    ps = np.array([r['pose_sens'] for r in res])
    out = dict(label=label, envs=len(res), pose_sens_max=float(ps.max()), pose_sens_median=float(np.median(ps)),
               envs_needing_a_level=int((C.K * ps > 1e-4).sum()), envs_with_pose_sens_above_1e_5=int((ps > 1e-5).sum()),
               envs_in_contact=int(sum(r['contacts'] > 0 for r in res)), worst=sorted(res, key=lambda r: -r['pose_sens'])[:6])
    print(json.dumps({k: v for k, v in out.items() if k != 'worst'}))
    return out


def from_dump(path, out_path):
    d = np.load(path)
    base = ModelBlob.load('bed_bathing_sawyer')
    f = base.obs_dim_robot - 1
    fw = C.float_words(base)
    runs = []
    for label, params in (('default', {}), ('manifold', {'MANIFOLD': 1.0}), ('warmstart 0.85', {'WARMSTART': 0.85}), ('friction along the fixed tangent (FRIC_EPS = 1e30)', {'FRIC_EPS': 1e30}),
                          ('two friction directions', {'FRICTION_DIRS': 2.0}), ('two friction directions, fixed tangents', {'FRICTION_DIRS': 2.0, 'FRIC_EPS': 1e30})):
        b = base
        for k, v in params.items():
            b = b.set_param(k, v)
        o = oracle_lib.Oracle(b)
        S, dev = [], []
        C.IN_SENSITIVITY[0] = True
        for j in range(len(d['picks'])):
            def run(st):
                o.forget_warm(); st = st.copy(); oo, r, dn, inf = o.step(st, d['action'][j]); return np.asarray(oo, dtype=np.float64)
            o0 = run(d['state'][j]); rng = np.random.RandomState(j); m = 0.0
            for _ in range(4):
                sp = d['state'][j].copy(); sp[fw] = C._perturb_f32(sp[fw], rng); m = max(m, float(np.delete(np.abs(run(sp) - o0), [f]).max()))
            S.append(m)
            if not params:
                dev.append(float(np.delete(np.abs(d['obs'][j] - o0), [f]).max()))
        C.IN_SENSITIVITY[0] = False
        S = np.array(S)
        r = dict(label=label, envs=len(S), pose_sens_max=float(S.max()), envs_flagged=int((C.K * S > 1e-4).sum()), flagged_picks=[int(x) for x in np.flatnonzero(C.K * S > 1e-4)],
                 flagged_pose_sens=[float('%.3g' % x) for x in S[C.K * S > 1e-4]])
        if not params:
            r['device_pose_deviation_of_the_flagged'] = [float('%.3g' % dev[x]) for x in np.flatnonzero(C.K * S > 1e-4)]
        print(json.dumps(r)); runs.append(r)
    out = dict(source=path, what='1-ulp sensitivity (4 trials) of the pose block of the observation after ONE step, f64 oracle, the 64 environments compared at the bench size on the GPU',
               K=C.K, tolerance=1e-4, runs=runs)
    if out_path:
        json.dump(out, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    if '--from' in sys.argv:
        from_dump(sys.argv[sys.argv.index('--from') + 1], arg('--out', ''))
        sys.exit(0)
    n, steps = arg('--envs', 48), arg('--steps', 20)
    from assistive_gym_amd.host.reset_bed import make_states
    base = ModelBlob.load('bed_bathing_sawyer')
    states = make_states(base, n, seed=2303)[0]
    out = dict(config='BedBathingSawyer-v1, random policy, %d steps, then the 1-ulp sensitivity of one more step (4 trials), f64 oracle' % steps, K=C.K, runs=[])
    for label, params in (('default', {}), ('manifold', {'MANIFOLD': 1.0}), ('warmstart 0.85', {'WARMSTART': 0.85}), ('manifold + warmstart 0.85', {'MANIFOLD': 1.0, 'WARMSTART': 0.85})):
        b = base
        for k, v in params.items():
            b = b.set_param(k, v)
        out['runs'].append(rollout_and_measure(b, states, steps, 4, label))
    path = arg('--out', '')
    if path:
        json.dump(out, open(path, 'w'), indent=1)
