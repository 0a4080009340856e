"""Consumer of the parity protocol's reference dumps.  `tools/pybullet_dump.py`, run
where the reference's PyBullet fork exists, records per step the full state (in this repo's state-record
layout), the action and the reference's observation / reward / done / total_force_on_human.  Here every
recorded state is injected, the recorded action applied, and the outputs compared (1e-3 relative, the
bound BASELINE.json's north_star names; absolute floor 1e-4).

No such dump can be produced in the build container (no pybullet): until a file tests/golden/pybullet_dump_*.npz is committed parity is
unpinned.  What IS committed are REHEARSAL files, tests/golden/bridge_dump_*.npz: the same tool run with --bridge on the
fake `pybullet` of tests/refbridge -- the reference's own env classes and step() on the CPU oracle's physics, for the five tasks the tool's
capture covers.  They are NOT PyBullet data (the oracle agrees with them by construction, to rounding); they keep the whole pipeline -- the
tool's capture between steps, the file format, both consumers below, on the oracle and through the C ABI on the GPU -- exercised end to end,
so that a real dump is a file drop."""
import glob
import os
import sys

import numpy as np
import pytest

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DUMPS = sorted(glob.glob(os.path.join(_G, 'pybullet_dump_*.npz'))) + sorted(glob.glob(os.path.join(_G, 'bridge_dump_*.npz')))
REL, FLOOR = 1e-3, 1e-4


def test_rehearsal_dumps_cover_the_tasks_and_say_what_they_are():
    names = [os.path.basename(p) for p in DUMPS if os.path.basename(p).startswith('bridge_dump_')]
    assert len(names) >= 5
    for p in DUMPS:
        d = np.load(p, allow_pickle=False)
        if os.path.basename(p).startswith('bridge_dump_'):
            assert 'NOT PyBullet' in str(d['source'])


def _check(name, got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), FLOOR / REL)
    assert err.max() < REL, '%s deviates from the reference dump: max rel. error %.3g' % (name, err.max())


def _load(path, blob):
    d = np.load(path, allow_pickle=False)
    if 'model' in d.files and str(d['model']) != 'feeding_jaco':          # a dump of another environment (tools/pybullet_dump.py --env: any task, any robot)
        from assistive_gym_amd.blob import ModelBlob
        blob = ModelBlob.load(str(d['model']))
    # what a dump depends on is the STATE RECORD layout.  Blob versions with the same record layout as the current one: 16 = 15 + one PARAMS entry
    # (AGX_P_SOLVE_WIDE, a device-only switch; include/agx_blob.h) -- the committed bridge rehearsals were recorded at 15
    assert int(d['blob_version']) in (blob.h['VERSION'], 15), 'dump was recorded for another blob version'
    assert d['states'].shape[1] == blob.state_words and len(d['states']) == len(d['actions']) + 1
    return d, blob


@pytest.mark.skipif(not DUMPS, reason='no PyBullet reference dump committed (tools/pybullet_dump.py needs the reference stack)')
@pytest.mark.parametrize('path', DUMPS)
def test_oracle_matches_reference_dump(path, blob, oracle):
    d, blob = _load(path, blob)
    if blob.words is not oracle.blob.words:
        from oracle_lib import Oracle
        oracle = Oracle(blob)
    has_cloth = 'cloth' in d.files and blob.task_kind == 3
    has_water = 'cloth' in d.files and blob.task_kind == 5
    import conditioning as C
    for k in range(len(d['actions'])):
        s = d['states'][k].copy()
        if has_water:                          # a Drinking dump carries the 64 water particles of every step (positions and velocities)
            obs, rew, done, info = oracle.step_cloth(s, d['cloth'][k].copy(), d['actions'][k])
            _check('oracle obs @%d' % k, obs, d['obs'][k]); _check('oracle reward @%d' % k, rew, d['reward'][k])
            _check('oracle force @%d' % k, info[0], d['total_force_on_human'][k])
            assert int(info[1]) == int(d['task_success'][k])
        elif has_cloth:                        # a Dressing dump carries the garment of every step (node velocities are not in the fork's API: zero)
            obs, rew, done, info = oracle.step_cloth(s, d['cloth'][k].copy(), d['actions'][k])
            # The cloth-force term (dressing.py:35-43; observation word 23, total_force_on_human, and reward through C_d = 0.01 per newton) is a
            # sum over hundreds of node contacts that switch on and off at a margin shell: the float32 rounding of the recorded garment alone
            # moves it by percent (the bridge rehearsal: up to 6 %, same oracle on both sides).  Judged against the oracle's own spread under a
            # 1e-6 m perturbation of the garment, as tests/test_gpu_dressing.py does; everything else at the contract tolerance.
            sens = C.ulp_sensitivity(blob, oracle, d['states'][k], d['actions'][k], cloth=d['cloth'][k], trials=3, seed=k, cloth_eps=1e-6)
            fdev = abs(float(obs[23]) - float(d['obs'][k][23]))
            ok, lim = C.check(fdev, REL * max(1.0, abs(d['obs'][k][23])), ulp=lambda: 4.0 * sens['obs'][23], uncapped=True)
            assert ok, ('cloth force sum @%d' % k, obs[23], d['obs'][k][23], lim)
            keep = np.arange(len(obs)) != 23
            _check('oracle obs @%d' % k, obs[keep], d['obs'][k][keep])
            assert abs(rew - d['reward'][k]) <= REL * max(1.0, abs(d['reward'][k])) + 0.01 * fdev, ('reward @%d' % k, rew, d['reward'][k])
            assert abs(info[0] - d['total_force_on_human'][k]) <= REL * max(1.0, abs(d['total_force_on_human'][k])) + fdev * 1.001 + 1e-6
        else:
            obs, rew, done, info = oracle.step(s, d['actions'][k])
            _check('oracle obs @%d' % k, obs, d['obs'][k]); _check('oracle reward @%d' % k, rew, d['reward'][k])
            _check('oracle force @%d' % k, info[0], d['total_force_on_human'][k])
        assert bool(done) == bool(d['done'][k])


@pytest.mark.gpu
@pytest.mark.skipif(not DUMPS, reason='no PyBullet reference dump committed (tools/pybullet_dump.py needs the reference stack)')
@pytest.mark.parametrize('path', DUMPS)
def test_stepper_matches_reference_dump(path, blob):
    from assistive_gym_amd import libagx
    from assistive_gym_amd.libagx import Stepper
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()
    d, blob = _load(path, blob)
    T = len(d['actions'])
    st = Stepper(blob, T)                     # step k of the episode runs in environment slot k
    st.set_state(d['states'][:T])
    if 'cloth' in d.files:
        st.set_cloth(d['cloth'][:T])
    obs, rew, done, info = st.step_host(d['actions'])
    import conditioning as C
    f = blob.obs_dim_robot - 1
    garment = 'cloth' in d.files and blob.task_kind == 3      # (a Drinking dump's `cloth` is the water: no cloth-force term)
    for k in range(T):
        pose = np.delete(np.arange(blob.obs_dim), f)
        if garment:
            pose = pose[pose != 23]              # the cloth-force term of the dressing observation: judged in tests/test_gpu_dressing.py against the oracle's own spread
        _check('obs @%d' % k, obs[k][pose], d['obs'][k][pose])
        # forces of a float32 pipeline carry the absolute floor of tests/conditioning.py (a contact is a spring of ~10^4 N/m in a gap known to ~1e-6 m)
        if not garment:                          # (dressing: this column IS the cloth-force sum, judged below)
            ok, lim = C.check(abs(obs[k, f] - d['obs'][k][f]), REL * max(1.0, abs(d['obs'][k][f])), C.force_floor(blob))
            assert ok, ('tool force @%d' % k, obs[k, f], d['obs'][k][f])
        cf = 0.0
        if garment:                             # the cloth-force term: see test_oracle_matches_reference_dump
            from oracle_lib import Oracle
            sens = C.ulp_sensitivity(blob, Oracle(blob), d['states'][k], d['actions'][k], cloth=d['cloth'][k], trials=3, seed=k, cloth_eps=1e-6)
            cf = abs(float(obs[k, 23]) - float(d['obs'][k][23]))
            ok, lim = C.check(cf, REL * max(1.0, abs(d['obs'][k][23])), ulp=lambda: 4.0 * sens['obs'][23], uncapped=True)
            assert ok, ('cloth force sum @%d' % k, obs[k, 23], d['obs'][k][23], lim)
        ok, lim = C.check(max(0.0, abs(info[k, 0] - d['total_force_on_human'][k]) - 1.001 * cf), REL * max(1.0, abs(d['total_force_on_human'][k])), C.force_floor(blob))
        assert ok, ('total_force_on_human @%d' % k, info[k, 0], d['total_force_on_human'][k])
        slack = 0.06 * C.force_floor(blob) + 0.01 * cf
        assert abs(rew[k] - d['reward'][k]) <= REL * max(1.0, abs(d['reward'][k])) + slack, ('reward @%d' % k, rew[k], d['reward'][k])
        assert bool(done[k]) == bool(d['done'][k])
    st.close()


# ---- the dump tool's state capture (tests/refbridge/capture.py, used by tools/pybullet_dump.py) against the bridge: capture(adopt(state))
# must give `state` back -- for every task, through the same PyBullet calls the tool makes where the real PyBullet exists
def _bridge_cases():
    import refbridge
    if not refbridge.available():
        return []
    import refcases
    want = ('feeding_jaco_tremor', 'feeding_food_events', 'feeding_coop_tremor', 'bed_wiping', 'bed_coop_rollback', 'scratch_itch_pr2_coop_scratching',
            'scratch_itch_jaco', 'arm_manipulation_sawyer_lifting', 'arm_manipulation_pr2', 'dressing_on_forearm', 'dressing_coop_sleeve',
            'feeding_stretch_step1', 'bed_bathing_stretch_step1', 'drinking_none_step0', 'drinking_spilling', 'drinking_at_the_mouth')
    out, seen = [], set()
    for c in refcases.build_cases():
        key = next((w for w in want if c['name'].startswith(w)), None)
        if key and key not in seen:
            seen.add(key); out.append(c)
    return out


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference checkout (the build container has it, the GPU box does not)')
def test_capture_inverts_adopt_on_the_bridge():
    import refbridge
    import refcases
    from refbridge import capture as cap
    from assistive_gym_amd.model import compiler as L
    cases = _bridge_cases()
    assert len(cases) >= 13 and any(c['model'].startswith('drinking') for c in cases)
    refbridge.install()
    p = sys.modules['pybullet']
    for c in cases:
        blob = refcases.variant_blob(c['model'], c['coop'], c['variant'])
        env, w = refbridge.adopt(blob, c['state'], c['cloth'])
        task = cap.TASK_OF_KIND[blob.task_kind]
        initial = {}
        if task == 'feeding':                       # adopt() lists only the particles that are still alive: the creation order comes from the body ids
            class _F:
                def __init__(self, body): self.body = body
                def __eq__(self, o): return getattr(o, 'body', None) == self.body
                def __hash__(self): return hash(self.body)
            initial['foods'] = [_F(refbridge.FOOD0 + k) for k in range(blob.nfood)]
            env.foods = [_F(f.body) for f in env.foods]; env.foods_active = [_F(f.body) for f in env.foods_active]
            env.bowl = _F(refbridge.BOWL)           # (created by reset() in the reference: furniture.py:32-34)
        if task == 'drinking':                      # likewise the water particles (drinking.py:160-171)
            class _W:
                def __init__(self, body): self.body = body
                def __eq__(self, o): return getattr(o, 'body', None) == self.body
                def __hash__(self): return hash(self.body)
            initial['waters'] = [_W(refbridge.WATER0 + k) for k in range(w.nwater)]
            env.waters = [_W(f.body) for f in env.waters]; env.waters_active = [_W(f.body) for f in env.waters_active]
        if task == 'bed_bathing':                   # likewise the wiping targets: marker ids in creation order
            ids = sorted(m for m in w.markers if m >= w.first_target_marker)
            nt = sum(int(x) for x in blob.task_i_n('NT', 4)[2 * w.gender:2 * w.gender + 2])
            initial['targets'] = ids[:nt]
        cl = None if c['cloth'] is None else np.zeros_like(c['cloth'])
        got = cap.capture(env, blob, p, initial, cloth_out=cl)
        a, b = blob.view(c['state'].reshape(1, -1).copy()), blob.view(got.reshape(1, -1))
        def pose_dev(x, y):                          # [..., 7+]: position, quaternion (q and -q are the same rotation), the rest
            x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
            dq = np.minimum(np.abs(x[..., 3:7] - y[..., 3:7]).max(axis=-1), np.abs(x[..., 3:7] + y[..., 3:7]).max(axis=-1))
            rest = np.abs(np.delete(x, [3, 4, 5, 6], axis=-1) - np.delete(y, [3, 4, 5, 6], axis=-1)).max(axis=-1)
            return float(np.maximum(dq, rest).max())
        nv = blob.h['BASE_LINK'] if blob.h['BASE_LINK'] > 0 else 0
        if nv:
            # a robot on a floating base (Stretch): the capture anchors the six virtual joints at the CURRENT base pose (angles zero, the base
            # twist in their rates) -- another record of the same physical state: the link frames agree, and so does what one env.step()
            # of the oracle makes of both
            from oracle_lib import Oracle
            o = Oracle(blob)
            pa, Ra = o.fk(c['state'].copy()); pb, Rb = o.fk(got.copy())
            assert np.abs(pa[nv - 1:] - pb[nv - 1:]).max() < 2e-6 and np.abs(Ra[nv - 1:] - Rb[nv - 1:]).max() < 2e-6, c['name']      # from the base link on (the virtual links before it are bookkeeping)
            s1, s2 = c['state'].copy(), got.copy()
            o1, o2 = o.step(s1, c['action']), o.step(s2, c['action'])
            assert np.abs(o1[0] - o2[0]).max() < 2e-5 and abs(o1[1] - o2[1]) < 2e-5, c['name']
            pa, _ = o.fk(s1); pb, _ = o.fk(s2)
            assert np.abs(pa[nv - 1:] - pb[nv - 1:]).max() < 2e-5, c['name']
            assert np.all(b['q'][0, :nv] == 0)
        for k in ('q', 'qd', 'tremor', 'tremor_target'):
            assert np.abs(a[k].astype(np.float64)[0, nv:] - b[k].astype(np.float64)[0, nv:]).max() < 2e-6 if k in ('q', 'qd') else np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() < 2e-6, (c['name'], k)
        assert (nv or pose_dev(a['base'][0], b['base'][0]) < 2e-6) and pose_dev(a['human'][0], b['human'][0]) < 2e-6, c['name']
        near = np.abs(a['free'][0][:, :3]).max(axis=1) < 500 if blob.nfree else np.zeros(0, bool)          # eaten particles were teleported away
        assert not near.any() or pose_dev(a['free'][0][near], b['free'][0][near]) < 2e-6, c['name']
        for k in ('gender', 'iteration', 'food_alive', 'food_active', 'task_success', 'frozen'):
            assert int(a[k][0]) == int(b[k][0]), (c['name'], k, int(a[k][0]), int(b[k][0]))
        assert abs(float(a['limit_scale'][0]) - float(b['limit_scale'][0])) < 1e-6 and abs(float(a['plane_friction'][0]) - float(b['plane_friction'][0])) < 1e-6
        nr = blob.nrobot
        passive = [d for d in range(nr) if blob.robot_i(d, 'ACT') < 0]
        assert np.abs(a['qt'][0, passive] - b['qt'][0, passive]).max() < 2e-6 if passive else True, c['name']
        ta, tb = a['task'][0].view(np.float32), b['task'][0].view(np.float32)
        if task == 'bed_bathing':
            assert np.array_equal(a['task'][0][:6], b['task'][0][:6]), c['name']                               # the surviving targets
        if task == 'scratch_itch':
            assert np.abs(ta[:3] - tb[:3]).max() < 1e-6 and a['task'][0][3] == b['task'][0][3] and np.abs(ta[12:15] - tb[12:15]).max() < 1e-6
        if task == 'dressing':
            assert abs(ta[L.DR['BEST']] - tb[L.DR['BEST']]) < 1e-6 and np.abs(cl[0] - c['cloth'][0]).max() < 2e-6
        if task == 'drinking':
            assert np.array_equal(a['task'][0][:4], b['task'][0][:4]), c['name']                               # alive / active particle masks
            assert np.abs(a['target'][0] - b['target'][0]).max() < 1e-6 and int(a['total_food'][0]) == int(b['total_food'][0])
            here = np.abs(c['cloth'][0]).max(axis=1) < 500                                                  # (drunk particles were teleported away)
            assert np.abs(cl[0][here] - c['cloth'][0][here]).max() < 2e-6 and np.abs(cl[1][here] - c['cloth'][1][here]).max() < 2e-6, c['name']
        if blob.task_i('ARM_LIMIT_ON'):
            assert a['task'][0][10] == b['task'][0][10] and np.abs(ta[6:10] - tb[6:10]).max() < 1e-6
        w.close()


def test_convention_search_recovers_the_conventions_of_a_synthetic_dump(tmp_path):
    """tools/convention_search.py on a dump whose "reference" is the oracle itself under a HIDDEN pair of conventions (warm start + second
    friction direction) -- the pad of BedBathingSawyer pressed on the arm, 10 steps with every state recorded: the search must rank exactly that
    combination first with zero deviation, and the default conventions must be told apart from it."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import convention_search as cs
    from bench import wiping_pool
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    b0 = ModelBlob.load('bed_bathing_sawyer')
    hidden = b0.set_param('WARMSTART', 0.85).set_param('FRICTION_DIRS', 2.0)
    o = Oracle(hidden); o.forget_warm()
    s = wiping_pool(b0, 4, 6006)[3].copy(); b0.view(s[None])['iteration'][0] = 0
    rng = np.random.RandomState(2)
    states, acts, obs, rew, done, force = [s.copy()], [], [], [], [], []
    for k in range(10):
        a = (rng.uniform(-1, 1, b0.act_dim) * 0.15).astype(np.float32)
        ob, r, dn, info = o.step(s, a)
        acts.append(a); obs.append(ob); rew.append(r); done.append(dn); force.append(info[0]); states.append(s.copy())
    o.forget_warm()
    path = str(tmp_path / 'pybullet_dump_synthetic.npz')
    np.savez(path, model='bed_bathing_sawyer', blob_version=b0.h['VERSION'], states=np.array(states), actions=np.array(acts), obs=np.array(obs),
             reward=np.array(rew), done=np.array(done), total_force_on_human=np.array(force))
    res = cs.search(path)
    best, r = res[0]
    assert best == (True, True, False, False, False), best
    assert max(r['obs'], r['reward'], r['force']) < 1e-9 and r['done_mismatches'] == 0
    default = [x for c, x in res if not any(c)][0]
    assert max(default['obs'], default['reward'], default['force']) > 1e-4, default
