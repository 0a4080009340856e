"""-m gpu: oracle parity AT THE BENCH SIZE for every BASELINE configuration (VERDICT r4 missing 3).  4096 environments in one handle -- the
chunk streams, the pool draws, the occupancy the bench line is measured at -- are rolled forward under a random policy; then ONE more step from
the states as they are, and a spread of the 4096 environments is compared with the CPU oracle one by one: observation, reward, the forces the
reward is made of.  Environments whose borderline contact candidates come out differently in float32 and float64 ("flips") are compared like all
others.  Quantities beyond the contract tolerance (north_star: 1e-3 relative on forces and rewards) go through tests/conditioning.py and are
counted in the run's tally.  PARITY UNPINNED vs PyBullet: the oracle is this repository's restatement."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gpu():
    from assistive_gym_amd import libagx
    if libagx.load().agx_device_count() <= 0:
        __import__('conftest').no_gpu()


def _compare(blob, oracle, before, act, obs, rew, info, picks, label, cloth=None, pose_tol=1e-4):
    """picked environments one by one.  Force entries of the observation (the tool force; co-op: the two force entries of the human's half),
    total_force_on_human, the robot's and the tool's force on the person (info 0, 2, 3) and the reward: 1e-3 relative, then the force floor, then the oracle's own 1-ulp
    and 1e-6 sensitivities (counted); everything else of the observation: pose_tol absolute."""
    import conditioning as C
    f = blob.obs_dim_robot - 1
    fcols = [f] + ([blob.obs_dim - 2, blob.obs_dim - 1] if blob.is_coop else [])
    ff = C.force_floor(blob)
    worst = dict(pose=0.0, reward=0.0, force=0.0)
    flips, judged, flagged = 0, 0, 0
    und0 = C.LAST_UNDETERMINED[0]
    for i in picks:
        s = before[i].copy()
        c = None if cloth is None else cloth[i].copy()
        o_obs, o_rew, o_done, o_info = oracle.step(s, act[i]) if c is None else oracle.step_cloth(s, c, act[i])
        flips += int(info[i, 6] != o_info[6])
        cache = {}

        def sens(eps, _i=i):
            if eps not in cache:
                cache[eps] = C.ulp_sensitivity(blob, oracle, before[_i], act[_i], cloth=None if cloth is None else cloth[_i], trials=4 if eps is None else 6, rel_eps=eps,
                                               cloth_eps=1e-6 if cloth is not None else None)
            return cache[eps]
        dev = np.abs(obs[i] - o_obs)
        pose = float(np.delete(dev, fcols).max())
        ok, lim = C.check(pose, pose_tol, ulp=lambda: C.K * float(np.delete(sens(None)['obs'], fcols).max()), step=lambda: C.K_STEP * float(np.delete(sens(C.STEP_EPS)['obs'], fcols).max()))
        assert ok, (label, i, 'pose', pose, lim)
        worst['pose'] = max(worst['pose'], pose)
        for k in fcols:
            ok, lim = C.check(dev[k], 1e-3 * max(1.0, abs(o_obs[k])), ff, ulp=lambda: C.K * sens(None)['obs'][k], step=lambda: C.K_STEP * sens(C.STEP_EPS)['obs'][k])
            assert ok, (label, i, 'obs force', k, obs[i, k], o_obs[k], lim)
            worst['force'] = max(worst['force'], dev[k] / max(1.0, abs(o_obs[k])))
        for q in (0, 2, 3):                           # total_force_on_human, the robot's force on the person, the tool's
            d = abs(float(info[i, q]) - float(o_info[q]))
            ok, lim = C.check(d, 1e-3 * max(1.0, abs(o_info[q])), ff, ulp=lambda: C.K * sens(None)['info'][q], step=lambda: C.K_STEP * sens(C.STEP_EPS)['info'][q])
            assert ok, (label, i, 'info', q, info[i, q], o_info[q], lim)
            worst['force'] = max(worst['force'], d / max(1.0, abs(o_info[q])))
        d = abs(float(rew[i]) - o_rew)
        # the reward carries the forces with the task's weights (at most 0.06 per newton, config.ini): its floor is that share of the force floor
        ok, lim = C.check(d, 1e-3 * max(1.0, abs(o_rew)), 0.06 * ff, ulp=lambda: C.K * sens(None)['reward'], step=lambda: C.K_STEP * sens(C.STEP_EPS)['reward'])
        assert ok, (label, i, 'reward', rew[i], o_rew, lim)
        worst['reward'] = max(worst['reward'], d / max(1.0, abs(o_rew)))
        assert info[i, 1] == o_info[1], (label, i, 'task_success', info[i, 1], o_info[1])
        judged += int(bool(cache))
        # device-independent: does the f64 oracle ALONE call this step ill-conditioned (K x its 1-ulp sensitivity above a compared quantity's tolerance)?
        if cloth is None:
            sn = sens(None)
            flagged += int(C.K * float(np.delete(sn['obs'], fcols).max()) > pose_tol or C.K * sn['reward'] > max(1e-3 * max(1.0, abs(o_rew)), 0.06 * ff) or
                           any(C.K * sn['info'][q] > max(1e-3 * max(1.0, abs(o_info[q])), ff) for q in (0, 2, 3)))
    undetermined = C.LAST_UNDETERMINED[0] - und0
    C.config_tally(label, len(picks), judged, undetermined, flagged)
    if os.environ.get('AGX_DUMP_BENCH_STATES'):        # the compared states for a CPU study (tests/diag/resting_contact_sensitivity.py --from <file>)
        os.makedirs(os.environ['AGX_DUMP_BENCH_STATES'], exist_ok=True)
        np.savez(os.path.join(os.environ['AGX_DUMP_BENCH_STATES'], 'bench_size_%s.npz' % label), picks=np.array(picks), state=before[picks], action=act[picks],
                 obs=obs[picks], reward=rew[picks], info=info[picks])
    print('%s: bench-size parity over %d of 4096 environments: worst pose %.2e, reward %.2e (relative), force %.2e (relative); contact-count flips %d (compared too); '
          'environments that needed a conditioning level %d (flagged ill-conditioned by the oracle alone: %d; beyond CAP x the tolerance, counted as undetermined: %d quantities)'
          % (label, len(picks), worst['pose'], worst['reward'], worst['force'], flips, judged, flagged, undetermined))
    # per configuration: at most 5 % of the compared environments may need a level -- or as many as the oracle alone flags, where that is more
    # (a property of the scene, not of the device); at most 2 % of the environments (at least one) may hold an undetermined quantity
    if cloth is None:
        assert judged <= max(int(C.MAX_JUDGED_PER_CONFIG * len(picks)), flagged), (label, judged, flagged)
        assert undetermined <= max(3, int(3 * C.MAX_UNDETERMINED * len(picks))), (label, undetermined)        # (quantities: one ill-conditioned environment holds up to three of them)
    return worst, flips, judged


def _rollout(env, steps, seed, scale=1.0):
    import torch
    n = env.n_envs
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    contacts = 0.0
    for k in range(steps):
        env.step((torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1) * scale)
        contacts += float(env.info[:, 6].mean())
    torch.cuda.synchronize()
    before = env.stepper.get_state()
    a = (torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1) * scale
    return before, a, contacts / max(steps, 1)


PICKS = [(i * 67) % 4096 for i in range(64)]


@pytest.mark.parametrize('config', ['config3_random', 'config3_wiping', 'config4_coop'])
def test_oracle_parity_at_bench_size(config):
    """BedBathingSawyer-v1 (random policy; the contact-rich wiping pool of bench.py) and ScratchItchPR2Human-v1 (co-op) at 4096 environments"""
    _gpu()
    import torch
    from assistive_gym_amd import vec_env
    from oracle_lib import Oracle
    n = 4096
    if config.startswith('config3'):
        env = vec_env.BedBathingSawyerVecEnv(n, pool_size=64, seed=2303)
        if config == 'config3_wiping':
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import wiping_pool
            env.set_pool(wiping_pool(env.blob, 64, 2303))
        steps, scale = (20, 1.0) if config == 'config3_random' else (5, 0.15)      # the wiping pool's episodes are 8 steps long: compare inside one
    else:
        env = vec_env.ScratchItchPR2HumanVecEnv(n, pool_size=64, seed=2404)
        steps, scale = 20, 1.0
    blob = env.blob
    oracle = Oracle(blob)
    env.reset()
    before, a, contacts = _rollout(env, steps, 7, scale)
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    obs, rew, info, a = obs.cpu().numpy(), rew.cpu().numpy(), info.cpu().numpy(), a.cpu().numpy()
    worst, flips, judged = _compare(blob, oracle, before, a, obs, rew, info, PICKS, config)
    print('%s: contacts per env step during the rollout %.2f' % (config, contacts))
    # (BedBathingSawyer under the random policy: an eighth of the compared environments has the arm on the mattress or the person at that step, where
    # one float32 ulp of the state moves the joint angles by 1e-3: judged against the oracle's own sensitivity and counted in the run's tally)
    assert flips <= 6
    env.close()


def test_oracle_parity_at_bench_size_dressing():
    """DressingBaxter-v1 at 4096 environments (4096 garments of 3,966 nodes): 16 environments against the oracle's 40 substeps + cloth solve.  The rigid
    part and the sleeve geometry at the contract tolerance; the cloth-force term (a sum over hundreds of node contacts that switch on and off
    at the margin shell) against the oracle's own spread under a 1e-6 m perturbation of the garment, as tests/test_gpu_dressing.py does."""
    _gpu()
    import torch
    import conditioning as C
    from assistive_gym_amd import vec_env
    from oracle_lib import Oracle
    n = 4096
    env = vec_env.DressingBaxterVecEnv(n, pool_size=32, seed=2505)
    blob = env.blob
    oracle = Oracle(blob)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(9)
    for k in range(6):
        env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
    torch.cuda.synchronize()
    picks = [(i * 67) % n for i in range(16)]
    before = env.stepper.get_state()
    cloth_t = env.stepper.cloth_tensor()
    before_c = {i: cloth_t[i].cpu().numpy().copy() for i in picks}
    a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    obs, rew, info, a = obs.cpu().numpy(), rew.cpu().numpy(), info.cpu().numpy(), a.cpu().numpy()
    rel_force, rel_sens = [], []
    for i in picks:
        s, c = before[i].copy(), before_c[i].copy()
        sens = C.ulp_sensitivity(blob, oracle, before[i], a[i], cloth=before_c[i], trials=2, seed=i, cloth_eps=1e-6)
        o_obs, o_rew, o_done, o_info = oracle.step_cloth(s, c, a[i])
        pose = float(np.abs(obs[i, :23] - o_obs[:23]).max())
        ok, lim = C.check(pose, 1e-4, ulp=lambda: C.K * float(sens['obs'][:23].max()))
        assert ok, (i, 'pose', pose, lim)
        rel_force.append(abs(obs[i, 23] - o_obs[23]) / max(1.0, abs(o_obs[23]))); rel_sens.append(sens['obs'][23] / max(1.0, abs(o_obs[23])))
        ok, lim = C.check(abs(info[i, 4] - o_info[4]), 1e-3 * max(1.0, abs(o_info[4])), ulp=lambda: C.K * sens['info'][4])      # reward_dressing
        assert ok, (i, 'reward_dressing', info[i, 4], o_info[4], lim)
        # reward = reward_dressing + C_d x cloth forces + preferences: beyond the cloth-force share (0.01 per newton of the difference) it is determined
        ok, lim = C.check(max(0.0, abs(rew[i] - o_rew) - 0.01 * abs(obs[i, 23] - o_obs[23])), 1e-3 * max(1.0, abs(o_rew)), ulp=lambda: C.K * sens['reward'])
        assert ok, (i, 'reward', rew[i], o_rew, lim)
    print('DressingBaxter at 4096: cloth_force_sum relative deviation, device vs oracle %s; oracle vs itself under 1e-6 m %s' % (np.round(rel_force, 4), np.round(rel_sens, 4)))
    assert np.median(rel_force) <= max(1e-3, 2.0 * np.median(rel_sens)) and max(rel_force) <= max(1e-3, 2.0 * max(rel_sens))
    env.close()


def test_plain_oracle_parity_config2_at_bench_size():
    """The approximations device and oracle SHARE (robot hulls decimated to <= 64 vertices, KEEP budgets on the food groups, the 1 mm solver slack, the 64 / 160 /
    2,040 contact / row / pair budgets, the no-op re-test rule) are invisible to every device-vs-oracle test.  Round 5 measured them on the CPU (default oracle vs
    a PLAIN oracle: tests/diag/approximation_budget.py); this is the same comparison ON THE HARDWARE: the product kernels with the product
    blob at 4096 environments against the PLAIN f64 oracle -- full hulls (tests/golden/feeding_jaco_plain.agxblob, written by tests/diag/make_plain_blob.py from the
    reference's assets), a row for every contact inside the break distance, budgets of 1024 / 4096 / 10^6, plain 50 sweeps -- for 64 environments over 20
    consecutive steps of a random-policy rollout, every step from the device's own state.  The CPU study found <= 1.5e-3 on one step in 600 (a food event);
    here: at most 1 % of the comparisons beyond 1e-3 relative on reward / total_force_on_human / tool force, no deviation beyond 5e-3 unless a food
    event (+20 / -5 / -1) fell on different sides, and at most 3 such events.
    One approximation PLAIN does NOT remove, and it is the one the first run of this test found (session r06g/h: 21 of 1,280 comparisons, ONE environment, reward
    off by 2e-2 ... 3e-2): the 42-direction penetration sampling that stands in for EPA when the CORES of two hulls overlap.  A Jaco link grazing the table's
    edge: its full hull cuts the corner by ~1 mm (one vertex 1.3 mm inside), the cores overlap, the sampled depth is 2.2 cm and the row pushes the arm away at
    1.1 m/s; the product's hull (64 vertices, <= 1.6 mm inside the full one) does not touch -- the device agrees with the product-convention oracle to 1.2e-4 on
    that environment, and is the one nearer to the truth.  Steps in which the PLAIN oracle took the sampling path (agxo_stat_core_overlaps) are therefore counted
    and compared separately (at most 3 % of the comparisons)."""
    _gpu()
    import torch
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    n, npick, nsteps = 4096, 64, 20
    env = vec_env.FeedingJacoVecEnv(n, pool_size=64, seed=2808)
    blob = env.blob
    plain = ModelBlob(np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeding_jaco_plain.agxblob'), dtype=np.uint32))
    assert plain.state_words == blob.state_words and plain.obs_dim == blob.obs_dim and plain.param('NOOP_RETEST') == 0 and plain.h['NVERT'] > 2 * blob.h['NVERT']
    op = Oracle(plain, plain=True)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(21)
    for k in range(15):
        env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
    pk = torch.from_numpy(np.array(PICKS)).to(env.device)
    f = blob.obs_dim_robot - 1
    import ctypes as Ct
    rel = dict(reward=[], total_force=[], tool_force=[])
    rel_sampled = []
    stat = (Ct.c_long * 2)()
    events, finite, sampled = 0, 0, 0
    keep = dict(state=[], action=[], reward=[], obs=[], info=[])
    for k in range(nsteps):
        torch.cuda.synchronize()
        st = env.stepper.state_tensor()[pk].cpu().numpy()
        a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        an, ob, rw, inf = a[pk].cpu().numpy(), obs[pk].cpu().numpy(), rew[pk].cpu().numpy(), info[pk].cpu().numpy()
        for key, val in (('state', st), ('action', an), ('reward', rw), ('obs', ob), ('info', inf)):
            keep[key].append(val.copy())
        for j in range(npick):
            s = st[j].copy()
            op.L.agxo_stat_core_overlaps(stat)
            po, pr, pd, pi = op.step(s, an[j])
            op.L.agxo_stat_core_overlaps(stat)
            if abs(float(rw[j]) - pr) > 0.5:                  # a food event (+20 eaten, -5 spilled, -1 touching the person) on one side only
                events += 1
                continue
            if stat[0] > 0:                                   # the PLAIN oracle estimated a penetration depth from 42 directions in this step (see the docstring)
                sampled += 1; rel_sampled.append(abs(float(rw[j]) - pr) / max(1.0, abs(pr)))
                continue
            finite += 1
            for key, x, y in (('reward', rw[j], pr), ('total_force', inf[j, 0], pi[0]), ('tool_force', ob[j, f], po[f])):
                rel[key].append(abs(float(x) - float(y)) / max(1.0, abs(float(y))))
    env.close()
    out = {key: dict(median=float(np.median(v)), p99=float(np.percentile(v, 99)), max=float(np.max(v)), frac_above_1e_3=float((np.array(v) > 1e-3).mean())) for key, v in rel.items()}
    out['steps_in_which_plain_sampled_a_penetration_depth'] = dict(count=sampled, reward_max=float(np.max(rel_sampled)) if rel_sampled else 0.0)
    print('config 2, device (product conventions) vs the PLAIN f64 oracle at the bench size: %d comparisons, %d with a food event on one side only;' % (finite, events), out)
    if os.environ.get('AGX_DUMP_BENCH_STATES'):
        import json
        os.makedirs(os.environ['AGX_DUMP_BENCH_STATES'], exist_ok=True)
        json.dump(dict(comparisons=finite, food_events_on_one_side_only=events, **out), open(os.path.join(os.environ['AGX_DUMP_BENCH_STATES'], 'plain_oracle_parity_config2.json'), 'w'), indent=1)
        np.savez(os.path.join(os.environ['AGX_DUMP_BENCH_STATES'], 'plain_oracle_parity_config2_states.npz'), **{key: np.stack(val) for key, val in keep.items()})
    assert events <= 3 and sampled <= 0.03 * npick * nsteps and finite >= npick * nsteps - 3 - sampled
    for key in rel:
        o = out[key]
        assert o['frac_above_1e_3'] <= 0.01 and o['max'] <= 5e-3 and o['median'] <= 1e-5, (key, o)


def _worker_init(paths):
    import sys
    for p in paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.pop('AGX_CONDITIONING_TALLY', None)


def test_cloth_force_distribution_at_bench_size():
    """The dressing task's cloth-force term (dressing.py:25,35-43: a sum over hundreds of node contacts, each in or out by thresholds) as a DISTRIBUTION
    (VERDICT r5 next 4a): environment by environment the device's sum deviates from the oracle's by 0.5 ... 9 %, the size of the oracle's own spread under a 1e-6 m
    perturbation of the garment -- which says nothing about a systematic error hiding inside that spread.  Here 1,024 of the 4,096 environments of a
    bench-size handle are compared over 5 consecutive steps (5,120 oracle steps, one process per host core):
      * the MEAN of the device's sums is within 1e-2 of the mean of the oracle's (a bias of the contact rule or of the force filter would show here);
      * the signed deviations are symmetric: a two-sided sign test does not reject P(device > oracle) = 1/2 at p = 0.01;
      * reported: the per-environment relative deviation (median, 90 %, max) and the rate at which a garment node is in contact on one side only."""
    _gpu()
    import multiprocessing as mp
    import torch
    from assistive_gym_amd import vec_env
    n, npick, nsteps = 4096, 1024, 5
    env = vec_env.DressingBaxterVecEnv(n, pool_size=32, seed=2707)
    blob = env.blob
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    for k in range(6):
        env.step(torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1)
    picks = np.arange(npick) * (n // npick)
    pk = torch.from_numpy(picks).to(env.device)
    f = blob.obs_dim_robot - 1
    nn = env.stepper.cloth_nodes()
    jobs, dev_sum, dev_nodes, dev_rew = [], [], [], []
    for k in range(nsteps):
        torch.cuda.synchronize()
        st = env.stepper.state_tensor()[pk].cpu().numpy(); cl = env.stepper.cloth_tensor()[pk].cpu().numpy()
        a = torch.rand((n, env.act_dim), device='cuda', generator=g) * 2 - 1
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        an = a[pk].cpu().numpy()
        rep = env.stepper.get_cloth_report()[picks][:, 20:].reshape(npick, nn, -1, 2)           # [env, node, slot, {height, |force|}]
        for j in range(npick):
            jobs.append(('dressing_baxter', st[j], cl[j], an[j]))
            dev_nodes.append(np.flatnonzero((rep[j, :, :, 1] >= 0).any(axis=1)).astype(np.int32))
        dev_sum += obs[pk, f].cpu().numpy().astype(np.float64).tolist(); dev_rew += rew[pk].cpu().numpy().astype(np.float64).tolist()
    env.close()
    here = os.path.dirname(os.path.abspath(__file__))
    import oracle_lib
    oracle_lib.lib()                                    # (built once, before the workers race for it)
    ctx = mp.get_context('spawn')
    with ctx.Pool(min(16, os.cpu_count() or 1), initializer=_worker_init, initargs=([here, os.path.dirname(here)],)) as pool:
        res = pool.map(oracle_lib.dressing_step_job, jobs, chunksize=8)
    dev = np.array(dev_sum); orc = np.array([r[0] for r in res]); orew = np.array([r[2] for r in res])
    d = dev - orc
    rel = np.abs(d) / np.maximum(1.0, np.abs(orc))
    pos, neg = int((d > 0).sum()), int((d < 0).sum())
    from scipy.stats import binomtest
    p_sign = float(binomtest(pos, pos + neg, 0.5).pvalue) if pos + neg else 1.0
    bias = float((dev.mean() - orc.mean()) / max(1.0, abs(orc.mean())))
    only_one, both = 0, 0
    for dn, r in zip(dev_nodes, res):
        a_, b_ = set(dn.tolist()), set(r[1].tolist())
        only_one += len(a_ ^ b_); both += len(a_ | b_)
    rew_beyond_share = np.maximum(0.0, np.abs(np.array(dev_rew) - orew) - 0.01 * np.abs(d))
    summary = dict(compared=len(dev), mean_device=float(dev.mean()), mean_oracle=float(orc.mean()), relative_bias_of_the_mean=bias, device_above=pos, device_below=neg, sign_test_p=p_sign,
                   relative_deviation_median=float(np.median(rel)), relative_deviation_p90=float(np.percentile(rel, 90)), relative_deviation_max=float(rel.max()),
                   node_contacts_on_one_side_only=only_one, node_contacts_on_either_side=both, node_contact_disagreement_rate=only_one / max(1, both),
                   environments_with_cloth_contact=int((orc > 0).sum()), reward_beyond_the_cloth_force_share_max=float(rew_beyond_share.max()))
    print('DressingBaxter cloth-force distribution at the bench size:', summary)
    if os.environ.get('AGX_DUMP_BENCH_STATES'):
        import json
        os.makedirs(os.environ['AGX_DUMP_BENCH_STATES'], exist_ok=True)
        json.dump(summary, open(os.path.join(os.environ['AGX_DUMP_BENCH_STATES'], 'cloth_force_distribution.json'), 'w'), indent=1)
    assert (orc > 0).sum() > 0.5 * len(orc), 'the workload has no cloth contact to speak of'
    assert abs(bias) <= 1e-2, summary
    assert p_sign >= 0.01, summary
    # (reported, not asserted: resting nodes sit exactly ON the margin shell the contact projection put them on -- which side of it a node is found on in the
    # last substep is decided by the last bit; what must not happen is a bias of the sum, tested above)


def test_oracle_parity_at_bench_size_dense_wiping():
    """BASELINE config 3 as BASELINE describes it ("dense tool-skin contact, PGS-heavy"): the scripted press-and-wipe policy of bench.py --workload
    dense (whole episodes with the pad on the arm, several contacts per substep) at 4096 environments; after 30 policy steps one more step from the
    states as they are, 64 environments against the oracle."""
    _gpu()
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import dense_wiping_pool, WipingPolicy
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.shard import pool_indices
    from oracle_lib import Oracle
    n, pool = 4096, 32
    env = vec_env.BedBathingSawyerVecEnv(n, pool_size=pool, seed=2606)
    blob = env.blob
    st, al, pr, touching = dense_wiping_pool(blob, pool, 0, seed=2606)
    env.set_pool(st)
    obs = env.reset()
    pol = WipingPolicy(al, pr, pool_indices(0, n, pool), blob.obs_dim_robot - 1, env.episode_len, env.device)
    contacts, forced = 0.0, 0.0
    for k in range(30):
        obs, _, _, info = env.step(pol(obs))
        contacts += float((info[:, 6] % 1000).mean()); forced += float((obs[:, blob.obs_dim_robot - 1] > 0).float().mean())
    torch.cuda.synchronize()
    before = env.stepper.get_state()
    a = pol(obs)
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    obs, rew, info, a = obs.cpu().numpy(), rew.cpu().numpy(), info.cpu().numpy(), a.cpu().numpy()
    worst, flips, judged = _compare(blob, Oracle(blob), before, a, obs, rew, info, PICKS, 'config3_dense')
    print('config3_dense: contacts per substep (last substep of a step, mean over 30 steps) %.2f; environments with a force on the pad %.0f %%; selection rollout %.0f %%'
          % (contacts / 30, 100 * forced / 30, 100 * touching))
    assert contacts / 30 >= 3.0 and forced / 30 > 0.5
    assert flips <= 8
    env.close()
