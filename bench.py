#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched stepper (BASELINE.json metric: FeedingJaco-v1, 4096 envs per MI355X;
`--task bedbathing` = BASELINE config 3, BedBathingSawyer-v1).

One "step" = one env.step() of every one of the 4096 lock-stepped environments of a GPU
(5 physics substeps of dt 0.02 + observation + reward), plus the auto-reset of
finished episodes from a device-resident pool.  Actions are a pre-generated random tape already in
HBM (random-policy rollout, the env_viewer loop of the reference).  With --gpus N every rank steps
its own 4096 environments (weak scaling) and the per-step observation all-gather over RCCL is
inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# The step is issued on 3 internal chunk streams + the caller's stream; with --gpus N the observation gather adds a side stream and RCCL its
# own.  HIP maps streams onto 4 hardware queues by default: a 5th stream shares a queue with a chunk stream and serialises two chunks
# (measured round 3: 4 chunks on 4 queues 324 k vs 466 k env-steps/s).  Must be in the environment before the HIP runtime initialises,
# i.e. before `import torch`; an explicit setting by the caller wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SHADER_CLOCK_HZ = 2.4e9   # same guide: max clock 2400 MHz; 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles
N_SIMD = 1024
SURVEY_8D_BYTES = {'feeding_jaco': 4253}     # SURVEY.md 8(d) "Algorithmic bytes per env-step (FeedingJaco)"
TASKS = {   # --task -> (model blob, VecEnv class, kernel-name suffix of the compiled variant, workload description)
    'feeding': ('feeding_jaco', 'FeedingJacoVecEnv', '', 'FeedingJaco-v1'),
    'feedingpanda': ('feeding_panda', 'FeedingPandaVecEnv', '', 'FeedingPanda-v1'),
    'bedbathing': ('bed_bathing_sawyer', 'BedBathingSawyerVecEnv', '_bb', 'BedBathingSawyer-v1'),
    'scratchitch': ('scratch_itch_pr2', 'ScratchItchPR2HumanVecEnv', '_si', 'ScratchItchPR2Human-v1 (co-op: 7 robot + 10 human actions)'),
    'armmanipulation': ('arm_manipulation_sawyer', 'ArmManipulationSawyerVecEnv', '_am', 'ArmManipulationSawyer-v1 (14 actions: the single arm listed twice)'),
    'feedingsawyer': ('feeding_sawyer', 'FeedingSawyerVecEnv', '_fl', 'FeedingSawyer-v1 (free-standing robot: feeding_l kernel variant, 320 colliders)'),
    'bedbathingpr2': ('bed_bathing_pr2', 'BedBathingPR2VecEnv', '_bbl', 'BedBathingPR2-v1 (bed_bathing_l kernel variant: 24 DoF)'),
    'scratchitchjaco': ('scratch_itch_jaco', 'ScratchItchJacoVecEnv', '_si', "ScratchItchJaco-v1 (the reference's default environment)"),
    'drinking': ('drinking_jaco', 'DrinkingJacoVecEnv', '_dk', 'DrinkingJaco-v1 (64 water particles per env, numSubSteps 4 / 10 solver sweeps: 20 internal substeps per step)'),
    'dressing': ('dressing_baxter', 'DressingBaxterVecEnv', '_dr', 'DressingBaxter-v1 (cloth of 3,966 nodes per env, numSubSteps 8: 40 internal substeps per step)'),
}


MULTI_GPU_CONFIGS = {      # world size -> (key in "configs", --task, timed steps): BASELINE.json configs[3] and configs[4]
    4: ('config4_ScratchItchPR2Human-v1_16384_envs_4gpu', 'scratchitch', 300),
    8: ('config5_DressingBaxter-v1_32768_envs_8gpu', 'dressing', 30),
}


def _cpu_worker(path, seed, n_steps, model='feeding_jaco'):
    """`bench.py --cpu-worker`: one host process stepping its share of the sample with the C oracle;
    prints "<env-steps> <seconds>"."""
    os.environ.pop('AGX_CONDITIONING_TALLY', None)      # (a timing leg, not a comparison: keep it out of a test session's conditioning tally)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    blob = ModelBlob.load(model[:-5]).coop() if model.endswith('+coop') else ModelBlob.load(model)
    o = Oracle(blob)
    init = np.load(path)
    st = init.copy()
    cloth0 = np.load(path[:-4] + '_cloth.npy') if os.path.exists(path[:-4] + '_cloth.npy') else None      # models with a garment
    cloth = cloth0.copy() if cloth0 is not None else None
    rng = np.random.RandomState(seed)
    t0 = time.perf_counter()
    for k in range(n_steps):
        if k and k % 200 == 0:
            st[:] = init            # episode end: back to a post-reset state, like the device-side auto-reset
            if cloth is not None:
                cloth[:] = cloth0
        a = rng.uniform(-1, 1, (len(st), blob.act_dim)).astype(np.float32)
        for i in range(len(st)):
            if cloth is not None:
                o.step_cloth(st[i], cloth[i], a[i])
            else:
                o.step(st[i], a[i])
    print(len(st) * n_steps, time.perf_counter() - t0)


def _usable_cores():
    """host cores this process may really use: the affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                             # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:                                         # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(blob, states, envs_per_core, n_steps, model='feeding_jaco', workload='FeedingJaco', cloth=None):
    """The CPU oracle (oracle/, plain C, f64) timed on a bounded sample of the same workload on ALL host
    cores of this box (one process per core, each stepping its own environments -- the reference's own
    scaling model, learn.py:26).  kind = "port": a restatement, NOT PyBullet."""
    import subprocess
    import tempfile
    cores = _usable_cores()
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for c in range(cores):
            path = os.path.join(tmp, 'cpu_%d.npy' % c)
            np.save(path, states[(c * envs_per_core + np.arange(envs_per_core)) % len(states)])
            if cloth is not None:
                np.save(path[:-4] + '_cloth.npy', cloth[(c * envs_per_core + np.arange(envs_per_core)) % len(states)])
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', path, str(c), str(n_steps), model],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        res = []
        for pr in procs:
            out, err = pr.communicate()
            if pr.returncode != 0 or len(out.split()) < 2:
                raise RuntimeError('cpu_baseline worker failed (rc %s): %s' % (pr.returncode, err.strip()[-400:]))
            steps, secs = out.split()[-2:]
            res.append((int(steps), float(secs)))
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    busy = max(r[1] for r in res)                       # slowest worker, excludes interpreter start-up
    single = np.mean([r[0] / r[1] for r in res])
    return dict(value=total / busy, unit='env-steps/s', cores=cores, kind='port',
                sample='%d processes x %d envs x %d steps of the same %s workload, C f64 oracle (not PyBullet); '
                       'per process %.0f env-steps/s; wall incl. start-up %.1f s' % (cores, envs_per_core, n_steps, workload, single, wall))


def wiping_pool(blob, n, seed):
    """BASELINE config 3 asks for 'dense tool-skin contact': post-reset states of BedBathingSawyer with the arm abducted and the wiping pad
    pressed 4 mm into the forearm / the upper arm at a random place (the robot base is translated so that the pad lands there; the
    reference's reset would have to be followed by a trained policy to get there).  Host code, pool generation only."""
    from assistive_gym_amd.host.reset_bed import make_states
    from assistive_gym_amd.model import xform as X
    n_want = n
    n = n + n // 2 + 4                                  # spare candidates: states whose arm is folded into itself are dropped below
    st, infos = make_states(blob, n, seed=seed, human_q_override={3: np.deg2rad(70)})
    rng = np.random.RandomState(seed)
    from assistive_gym_amd.model.human import HumanModel
    for i in range(n):
        s = st[i]
        v = blob.view(s.reshape(1, -1))
        g = int(v['gender'][0])
        hm = HumanModel('male' if g == 0 else 'female')
        nr = blob.nrobot
        # world frames of the right arm links from the state's static / dynamic human records: joint chain 0..9 hangs off the chest
        q = np.zeros(hm.n); q[:10] = v['q'][0, nr:nr + 10]
        base = v['human'][0, 0]
        pos, quat = hm.fk(base[:3].astype(np.float64), base[3:7].astype(np.float64), q)
        sh, el, wr = pos[5], pos[7], pos[9]
        fore = rng.rand() < 0.6
        p0, p1 = (el, wr) if fore else (sh, el)
        rad = hm.dims['forearm' if fore else 'upperarm'][0]
        want = p0 + rng.uniform(0.25, 0.75) * (p1 - p0) + np.array([0, 0, rad + 0.0025 - 0.004])
        fp, fq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
        bp, bq = X.compose(fp, fq, blob.free_f(0, 'REFPOS', 3), blob.free_f(0, 'REFQUAT', 4))
        pad, _ = X.compose(bp, bq, blob.task_f('TOOL_OBS_POS', 3), blob.task_f('TOOL_OBS_QUAT', 4))
        d = (want - pad).astype(np.float32)
        v['base'][0, :3] += d; v['free'][0, 0, :3] += d; v['free'][0, 0, 7:] = 0
        # the pad stays on the skin for a handful of steps under small random actions (no policy presses it down): episodes of this
        # workload are 8 steps long, i.e. every environment is put back onto the arm from the pool every 8 steps
        v['iteration'][0] = int(blob.task_f('EPISODE_LEN')) - 8
    # The host sampler's least-squares IK does not keep the arm out of itself (the reference's null-space IK does): a start with two robot links
    # 5 cm inside each other is not a wiping state, it is an explosion a few steps later (round 5: the stepper's own collision flags drop them;
    # without a GPU -- the emulator tests -- the candidates are taken as they come).
    from assistive_gym_amd import libagx
    keep = np.ones(n, dtype=bool)
    if libagx.load().agx_device_count() > 0:
        chk = libagx.Stepper(blob, n)
        chk.set_state(st)
        keep = (chk.check_collisions() & 2) == 0         # AGX_COLLIDE_SELF (include/agx.h): two robot links, or a link and the tool, more than 1 cm inside each other
        chk.close()
    idx = np.flatnonzero(keep)
    if len(idx) < n_want:                               # (never seen: a third of the candidates are spare)
        idx = np.concatenate([idx, np.flatnonzero(~keep)])
    return np.ascontiguousarray(st[idx[:n_want]])


def wiping_tape_vectors(blob, states):
    """A scripted press-and-wipe policy for the 'dense' workload of config 3 (BASELINE: "dense tool-skin contact, PGS-heavy"):
    per start state two joint-space directions -- `along`: the arm joints' motion that moves the pad along the limb it lies on, `press`: the one
    that pushes it into the skin -- from the numeric Jacobian of the pad position at the start pose (host/kin.py), damped least squares.
    The action of step t is g(F) x press + wipe_gain x along x square_wave(t) (WipingPolicy below): no learned policy.  The reference's take_step
    re-anchors the motor targets at the CURRENT angles every step (env.py:201-215), so a constant press does not wind up."""
    from assistive_gym_amd.host.kin import RobotKin
    from assistive_gym_amd.model import xform as X
    from assistive_gym_amd.model.human import HumanModel
    kin = RobotKin(blob)
    arm = kin.arm
    n = len(states)
    along, press = np.zeros((n, blob.act_dim), dtype=np.float32), np.zeros((n, blob.act_dim), dtype=np.float32)
    for i in range(n):
        v = blob.view(states[i].reshape(1, -1))
        q = v['q'][0, :blob.nrobot].astype(np.float64)
        bp, bq = v['base'][0, :3].astype(np.float64), v['base'][0, 3:7].astype(np.float64)

        def pad(qq):
            tp, tq = kin.tool_pose(bp, bq, qq)
            return X.compose(tp, tq, blob.task_f('TOOL_OBS_POS', 3), blob.task_f('TOOL_OBS_QUAT', 4))[0]
        p0 = pad(q)
        J = np.zeros((3, len(arm)))
        for k, d in enumerate(arm):
            dq = q.copy(); dq[d] += 1e-5
            J[:, k] = (pad(dq) - p0) / 1e-5
        hm = HumanModel('male' if int(v['gender'][0]) == 0 else 'female')
        hq = np.zeros(hm.n); hq[:10] = v['q'][0, blob.nrobot:blob.nrobot + 10]
        base = v['human'][0, 0]
        pos, _ = hm.fk(base[:3].astype(np.float64), base[3:7].astype(np.float64), hq)
        sh, el, wr = pos[5], pos[7], pos[9]
        # the limb the pad is nearest to, its axis, and the direction from the pad to the axis (= into the skin)
        best = None
        for a0, a1 in ((el, wr), (sh, el)):
            ax = (a1 - a0) / np.linalg.norm(a1 - a0)
            t = float(np.clip(np.dot(p0 - a0, ax), 0.0, np.linalg.norm(a1 - a0)))
            foot = a0 + t * ax
            dist = np.linalg.norm(p0 - foot)
            if best is None or dist < best[0]:
                best = (dist, ax, (foot - p0) / max(dist, 1e-9))
        Jp = J.T @ np.linalg.inv(J @ J.T + 1e-4 * np.eye(3))                      # damped pseudo-inverse
        for vec, out in ((best[1], along), (best[2], press)):
            dq = Jp @ vec
            dq = dq / max(np.abs(dq).max(), 1e-9)
            for k, d in enumerate(arm):
                out[i, kin.act[d]] = dq[k]
    return along, press


class WipingPolicy:
    """The scripted press-and-wipe policy on the device: action = press x g(F) + along x wipe_gain x square_wave(t), g(F) = clip(0.05 + kf (F_target - F),
    -0.05, 0.3) with F the tool force of the last observation -- press harder while the pad carries less than F_target newtons, ease off above.
    Five tensor operations per step on the observations that are already on the device."""

    def __init__(self, along, press, idx, fcol, episode_len, device, kf=0.1, f_target=2.0, wipe_gain=0.25, period=40):
        import torch
        self.torch = torch
        self.al = torch.from_numpy(along[idx]).to(device) * wipe_gain
        self.pr = torch.from_numpy(press[idx]).to(device)
        self.fcol, self.T, self.kf, self.ft, self.period, self.t = fcol, episode_len, kf, f_target, period, 0

    def __call__(self, obs):
        torch = self.torch
        wave = 1.0 if (self.t % self.T) % self.period < self.period // 2 else -1.0
        self.t += 1
        g = torch.clamp(0.05 + self.kf * (self.ft - obs[:, self.fcol]), -0.05, 0.3)
        return torch.clamp(self.pr * g[:, None] + self.al * wave, -1.0, 1.0).contiguous()


def dense_wiping_pool(blob, pool, device_index, seed=1001):
    """Start states for the dense workload: twice `pool` candidates of the wiping pool, every one rolled for a whole 200-step episode under the
    scripted policy on the device; the `pool` states on which the policy keeps the pad on the skin longest are kept (the rest slide off the limb
    within a few steps -- an open-loop direction computed at the start pose is all the policy knows about the arm).  -> (states, along, press)"""
    import torch
    from assistive_gym_amd import vec_env
    cand = wiping_pool(blob, 2 * pool, seed)
    blob.view(cand)['iteration'][:] = 0                      # whole episodes
    along, press = wiping_tape_vectors(blob, cand)
    m = len(cand)
    sel = vec_env.BedBathingSawyerVecEnv(m, device=device_index, seed=seed, pool_size=m, blob=blob)
    sel.set_pool(cand)
    obs = sel.reset()
    f = blob.obs_dim_robot - 1
    pol = WipingPolicy(along, press, np.arange(m), f, sel.episode_len, sel.device)
    touching = torch.zeros(m, device=sel.device)
    for k in range(sel.episode_len - 1):
        obs, _, _, _ = sel.step(pol(obs))
        touching += (obs[:, f] > 0).float()
    order = torch.argsort(touching, descending=True).cpu().numpy()[:pool]
    sel.close()
    return np.ascontiguousarray(cand[order]), along[order], press[order], float(touching[torch.from_numpy(order).to(touching.device)].mean().item()) / (sel.episode_len - 1)


class _DryEnv:
    """--dry-run: a stand-in for the batched environment on CPU tensors, so that the launch path of `bench.py --gpus N` -- rendezvous from the
    torchrun environment, sharding by rank, the per-step observation all-gather (shard.ObsGatherer), barrier, max-over-ranks timing, the one
    JSON line on rank 0 -- can be exercised without a GPU (tests/test_dist_gloo.py, backend gloo).  It steps nothing: libagx has no CPU path."""

    def __init__(self, n, act_dim, obs_dim, rank):
        import torch
        self.n_envs, self.act_dim, self.obs_dim, self.rank = n, act_dim, obs_dim, rank
        self.obs = torch.zeros((n, obs_dim)); self.info = torch.zeros((n, 8)); self.env_offset = 0
        self.reward = torch.zeros(n); self.done = torch.zeros(n, dtype=torch.uint8)

    def reset(self, env_offset=0):
        self.env_offset = env_offset

    def step(self, actions, obs_out=None):
        import torch
        o = self.obs if obs_out is None else obs_out
        o.zero_(); o[:, :self.act_dim] = actions; o[:, -1] = torch.arange(self.n_envs, dtype=torch.float32) + self.env_offset     # global env index: checked by the test
        self.obs = o
        g = torch.arange(self.n_envs, dtype=torch.float32) + self.env_offset
        self.reward = -g; self.done = (g.long() % 2).to(torch.uint8); self.info[:, 0] = 0.5 * g; self.info[:, 1] = 2.0 * g      # functions of the global index: the packed record is checked column by column
        return o, self.reward, self.done, self.info


def dry_run(args, rank, world):
    import torch
    import torch.distributed as dist
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.shard import BatchGatherer
    model, _, _, env_id = TASKS[args.task or 'feeding']
    blob = ModelBlob.load(model)
    if env_id.split()[0].endswith('Human-v1'):
        blob = blob.coop()
    n, K, W = args.envs_per_gpu, args.steps, args.warmup
    env = _DryEnv(n, blob.act_dim, blob.obs_dim, rank)
    env.reset(env_offset=rank * n)
    g = torch.Generator(); g.manual_seed(1001 + rank)
    tape = torch.rand((W + K, n, blob.act_dim), generator=g) * 2 - 1
    # the whole-batch record of a step: observation | reward | done | total_force_on_human | task_success, one all-gather
    gatherer = BatchGatherer(n, blob.obs_dim + 4, world, device=None)
    full = None
    for k in range(W + K):
        if k == W:
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        o, r, d, i = env.step(tape[k])
        gatherer.pack(k & 1, o, r, d, i); full = gatherer.submit(k & 1)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gi, od = torch.arange(world * n, dtype=torch.float32), blob.obs_dim
    # every rank holds the whole batch in global env order, every column of the record where it belongs
    ok = bool(torch.equal(full[:, od - 1], gi) and torch.equal(full[:, od], -gi) and torch.equal(full[:, od + 1], (gi.long() % 2).float()) and
              torch.equal(full[:, od + 2], 0.5 * gi) and torch.equal(full[:, od + 3], 2.0 * gi))
    return {'metric': 'env_steps_per_sec', 'value': world * n * K / float(t.item()), 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': float(t.item()) / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'DRY RUN of the launch path of %s (no stepping: libagx has no CPU path)' % env_id, 'envs_per_gpu': n, 'global_envs': world * n,
                       'parallelism': 'env-sharded x%d' % world, 'obs_allgather': world > 1, 'gather': 'torch.distributed (%s; the dry run has no device buffers for the C ABI collective)' % (args.backend or 'gloo'), 'gathered_record': 'obs | reward | done | total_force_on_human | task_success', 'gathered_in_global_order': ok, 'obs_dim': blob.obs_dim, 'act_dim': blob.act_dim},
            'dry_run': True}


def run_config(args, task, steps, warmup, rank, world, local_rank, distributed, cpu=True, workload=None, env_id_override=None):
    """one timed run of one configuration on this rank's GPU; returns the JSON-able dict on rank 0 (None elsewhere)"""
    import torch
    import torch.distributed as dist
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.shard import BatchGatherer
    model, env_cls, ksuffix, env_id = TASKS[task]
    pool = args.pool
    if env_id_override is not None:             # the kernel-name suffix is that of the variant agx_create picks: taken from the stepper below
        from assistive_gym_amd.envs import ENV_IDS
        cls = ENV_IDS[env_id_override.split(':')[-1]]
        model, env_cls, ksuffix, env_id = cls.model, None, None, env_id_override.split(':')[-1]
        task = 'dressing' if model.startswith('dressing') else 'drinking' if model.startswith('drinking') else env_id_override
    if pool is None:
        pool = 64 if task == 'dressing' or workload == 'dense' else 256       # (dense: every candidate start is rolled for an episode on the device first)
    n = args.envs_per_gpu
    blob_override = None
    if args.param:
        from assistive_gym_amd.blob import ModelBlob
        blob_override = ModelBlob.load(model)
        for kv in args.param:
            k, v = kv.split('=')
            blob_override = blob_override.set_param(k, float(v))
    if env_cls is None:
        env = vec_env.AssistiveVecEnv(n, device=local_rank, seed=1001, pool_size=pool, reset=args.reset, model=model, coop=env_id.endswith('Human-v1'), blob=blob_override,
                                      pool_refresh=getattr(args, 'pool_refresh', 0))
        ksuffix = {'feeding': '', 'feeding_l': '_fl', 'feeding_m': '_fm', 'bed_bathing': '_bb', 'bed_bathing_l': '_bbl', 'bed_bathing_m': '_bbm', 'scratch_itch': '_si', 'scratch_itch_m': '_sim', 'dressing': '_dr', 'dressing_l': '_drl', 'dressing_m': '_drm',
                   'arm_manipulation': '_am', 'arm_manipulation_l': '_aml', 'drinking': '_dk', 'drinking_l': '_dkl', 'drinking_m': '_dkm'}[env.stepper.variant()]
    else:
        env = getattr(vec_env, env_cls)(n, device=local_rank, seed=1001, pool_size=pool, reset=args.reset, blob=blob_override, pool_refresh=getattr(args, 'pool_refresh', 0))
    blob = env.blob                      # the co-op flavour where the task's BASELINE config is co-op
    action_scale = 1.0
    policy, dense_touching = None, None
    if workload == 'wiping':
        env.set_pool(wiping_pool(blob, pool, 1001))
        action_scale = 0.15
    if workload == 'dense':
        from assistive_gym_amd.shard import pool_indices
        st_d, al_d, pr_d, dense_touching = dense_wiping_pool(blob, pool, local_rank)
        env.set_pool(st_d)
        policy = WipingPolicy(al_d, pr_d, pool_indices(rank * n, n, pool), blob.obs_dim_robot - 1, env.episode_len, env.device)
    env.reset(env_offset=rank * n)
    K, W = steps, warmup
    g = torch.Generator(device='cuda'); g.manual_seed(1001 + rank)
    tape = (torch.rand((W + K, n, blob.act_dim), device='cuda', generator=g) * 2 - 1) * action_scale if policy is None else None
    # whole-batch collation: per step ONE record per environment -- observation | reward | done | total_force_on_human | task_success (SURVEY
    # 8e; packed on the device by agx_pack_step) -- all-gathered over RCCL on a side stream, overlapped with the next step (two buffers alternate).
    # --gather abi (the default): the collective of the C ABI (agx_comm_init_rank / agx_allgather, RCCL bound by libagx itself; the 128-byte id
    # travels over the torch.distributed group that the launcher's rendezvous set up); --gather torch: torch.distributed's all-gather.
    gatherer, gather_how, comm = None, None, None
    if distributed:
        gather_how = args.gather
        if gather_how == 'abi':
            # agx_comm_init_rank is a COLLECTIVE: a rank that cannot take part (RCCL not bindable by libagx, ...) must keep the others from
            # entering it, and a rank whose own init failed must take the others with it into the fallback -- otherwise some ranks sit in
            # ncclCommInitRank / agx_allgather while the rest call dist.all_gather: a hang, not a fallback.  Two votes over the
            # launcher's torch.distributed group: before the collective (can every rank bind RCCL and did rank 0 get an id?) and after it.
            from assistive_gym_amd import libagx
            err = None
            try:
                uid = libagx.comm_unique_id()            # (every rank: binds RCCL or raises; rank 0's id is the one that travels)
                box = [uid if rank == 0 else None]
            except Exception as e:
                err, box = str(e), [None]
            dist.broadcast_object_list(box, src=0)
            ok = torch.tensor([0 if (err or box[0] is None) else 1], device='cuda', dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()):
                try:
                    comm = libagx.comm_init_rank(local_rank, rank, world, box[0])
                except Exception as e:
                    err, comm = str(e), None
                ok = torch.tensor([0 if comm is None else 1], device='cuda', dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if not int(ok.item()) and comm is not None:
                    libagx.comm_destroy(comm); comm = None
            if comm is None:                             # (on every rank alike; the line says so)
                gather_how = 'torch (agx_comm_init_rank failed on %s: %s)' % ('this rank' if err else 'another rank', str(err)[:120])
        gatherer = BatchGatherer(n, blob.obs_dim + 4, world, device=torch.device('cuda', local_rank), force=args.force_gather, stepper=env.stepper, comm=comm)

    def one(k):
        env.step(tape[k] if policy is None else policy(env.obs))
        if distributed:
            gatherer.pack(k & 1, env.obs, env.reward, env.done, env.info)
            gatherer.submit(k & 1)

    # the contact sampling of the timed loop (torch remainder / cast / mean / add kernels and their allocations) runs in the warm-up too:
    # the first call of each loads its code object (tens of ms each) -- inside a 20-step timed window that was half of the time
    # (236 k in the driver's 20-step run vs 471 k over 2,000 steps)
    ncon_sum, ncon_buf = torch.zeros((), device='cuda', dtype=torch.float64), torch.zeros((), device='cuda', dtype=torch.float64)

    def sample_contacts():
        torch.mean((env.info[:, 6] % 1000).double(), dim=0, out=ncon_buf)
        ncon_sum.add_(ncon_buf)

    for k in range(W):
        one(k)
        sample_contacts()
    sample_contacts()
    ncon_sum.zero_()
    if distributed:
        gatherer.wait()
    stream = torch.cuda.current_stream().cuda_stream
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    overflow0 = env.stepper.overflow_count()
    env.stepper.profile_begin(stream)
    t0 = time.perf_counter()
    ncon_n = 0
    for k in range(W, W + K):
        one(k)
        if (k - W) % 16 == 0:            # contacts of the last substep, sampled (device-side sum, no host sync)
            sample_contacts(); ncon_n += 1
    kernel_ms = env.stepper.profile_end(stream)
    if distributed:
        gatherer.wait()                  # the last gathers are inside the timed region
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    in_order = None
    if distributed:
        # after the timed region: is the whole batch every rank holds in GLOBAL env order?  Rank r's own shard must sit bit for bit at rows
        # [r n, (r + 1) n) of its gathered buffer, and the per-shard checksums every rank computed of its OWN records must be the ones found
        # at the shards' places in this rank's gathered buffer (all ranks vote)
        last = (W + K - 1) & 1
        full, mine = gatherer.full[last], gatherer.local[last]
        sums = torch.zeros(world, device='cuda', dtype=torch.float64)
        sums[rank] = mine.double().sum()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        found = full.double().reshape(world, -1).sum(dim=1)
        ok = torch.equal(full[rank * n:(rank + 1) * n], mine) and bool(((found - sums).abs() <= 1e-9 * sums.abs().clamp(min=1.0)).all())
        okt = torch.tensor([1 if ok else 0], device='cuda', dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        in_order = bool(okt.item())
    overflow = env.stepper.overflow_count() - overflow0
    if comm is not None:
        from assistive_gym_amd import libagx
        torch.cuda.synchronize(); libagx.comm_destroy(comm)
    contacts = float(ncon_sum.item()) / max(1, ncon_n)
    # per-kernel launch durations (HIP events after every launch, on the chunk stream it is launched on),
    # measured after the timed region so that `value` is not perturbed by the extra events / host syncs.
    # The step is issued as a few independent chunks of environments on internal streams, so launches of
    # different chunks overlap: the durations are those of launches that share the GPU, exactly what a
    # kernel trace (rocprofv3 --kernel-trace) of this command reports.
    NT = 20
    kms, kcnt = np.zeros(3), np.zeros(3)
    has_cloth = env.stepper.cloth_nodes() > 0
    if has_cloth:
        # 40 build / solve pairs + the cloth kernel + finish per step: too many launches for the per-launch event table; the split by
        # kernel comes from the rocprofv3 kernel trace (profiles/); here the whole step is the unit
        kms[:] = [0, 0, kernel_ms / K]; kcnt[:] = [0, 0, env.stepper.n_chunks() if hasattr(env.stepper, 'n_chunks') else 1]
    else:
        for k in range(NT):
            act = tape[(W + k) % (W + K)] if policy is None else policy(env.obs)
            ms, cnt = env.stepper.step_timed(act, env.obs, env.reward, env.done, env.info, stream)
            kms += np.array(ms); kcnt += np.array(cnt)
        kms /= NT; kcnt /= NT
    out = None
    if rank == 0:
        total_steps = world * n * K
        value = total_steps / elapsed
        sw = blob.state_words
        # algorithmic HBM bytes per env-step: state record read + written once, action read, obs /
        # reward / done / info written (DESIGN.md "bytes per env-step")
        bytes_per_env_step = 2 * sw * 4 + blob.act_dim * 4 + blob.obs_dim * 4 + 4 + 1 + 8 * 4
        if has_cloth:                        # + the garment read and written once per env step: node positions and velocities
            bytes_per_env_step += 2 * 6 * env.stepper.cloth_nodes() * 4
        # `roofline.achieved` uses SURVEY 8d's per-unit figure where the survey gives one (FeedingJaco: 4,253 B per env-step, a 514-word
        # state incl. a 256-word contact warm-start cache this design does not keep); the bytes of the record this design really moves
        # (322 words: 2,741 B) stay beside it as `record_bytes_per_env_step`
        record_bytes_per_env_step = bytes_per_env_step
        if not blob.is_coop:
            bytes_per_env_step = SURVEY_8D_BYTES.get(model, bytes_per_env_step)
        names = [k + ksuffix for k in ('agx_build_kernel', 'agx_solve_kernel', 'agx_finish_kernel')]
        traffic_key = None
        if has_cloth:
            nsub = int(blob.param('FRAME_SKIP')) * max(1, blob.h['SIM_SUBSTEPS'])
            pk = 'agx_water_kernel' if task == 'drinking' else 'agx_cloth_kernel'
            names[2] = 'whole step (%d x [agx_build_kernel%s, agx_solve_kernel%s] + %s%s + agx_finish_kernel%s)' % (nsub, ksuffix, ksuffix, pk, ksuffix, ksuffix)
            traffic_key = pk + ksuffix
        dom = int(np.argmax(kms))
        # one launch of the dominant kernel advances the environments of one chunk by 1/frame_skip of an env-step
        launches = [int(round(x)) for x in kcnt]
        chunks = max(1, launches[2])
        envs_per_launch = n / chunks
        launch_ms = kms[dom] / launches[dom]
        units = n / launches[dom]            # env-steps advanced by one launch (= envs_per_launch / frame_skip for build / solve)
        achieved = bytes_per_env_step * units / (launch_ms * 1e-3) / 1e9
        # HBM traffic and VALU instruction counts of the same kernel from separate rocprofv3 --pmc passes (tools/pmc_workload.py,
        # reduced by tools/pmc_traffic.py to per-environment figures), scaled to the SAME launch size as the algorithmic bytes
        traffic, traffic_src, valu_frac, cloth_kernel = None, None, None, None
        tname = task if workload is None else task + '_' + workload
        for cand in ('r06_traffic_%s.json' % tname, 'r05_traffic_%s.json' % tname, 'r04_traffic_%s.json' % tname, 'r03_traffic_%s.json' % tname, 'r02_traffic_%s.json' % tname, 'r01_traffic.json' if task == 'feeding' else None):
            tpath = cand and os.path.join(ROOT, 'profiles', cand)
            if tpath and os.path.exists(tpath):
                tj = json.load(open(tpath))
                kj = tj.get('kernels', {}).get(traffic_key or names[dom])
                if kj:
                    traffic = kj['hbm_bytes_per_env_launch'] * envs_per_launch
                    traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s), per environment x %d environments per launch' % (cand, tj.get('correction', ''), envs_per_launch)
                    if 'valu_insts_per_env_launch' in kj:
                        # wave64 VALU instruction = 2 issue cycles on a SIMD-32 (guide, "Wave scheduling"); 1024 SIMDs
                        ms_for = kj.get('ms_per_launch', launch_ms) if traffic_key else launch_ms
                        valu_frac = kj['valu_insts_per_env_launch'] * envs_per_launch * 2.0 / (ms_for * 1e-3 * SHADER_CLOCK_HZ * N_SIMD)
                    if traffic_key:
                        cloth_kernel = dict(kernel=traffic_key, **{k: kj[k] for k in kj if k != 'hbm_bytes_per_env_launch'}, hbm_bytes_per_env_launch=kj['hbm_bytes_per_env_launch'])
                    break
        out = {
            'metric': 'env_steps_per_sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s, %d lockstep envs per MI355X, %s, 5 simulation steps per env step, %d PGS sweeps' % (env_id, n, 'random-policy rollout' if workload is None else ('scripted press-and-wipe policy on the device (tool force feedback), whole 200-step episodes from starts with the pad on the arm; the pad carries force in %.0f %% of the selection rollout' % (100 * dense_touching)) if workload == 'dense' else 'pad pressed onto the arm at every reset, 8-step episodes, small random actions (x%.2f)' % action_scale, int(blob.param('NITER'))),
                       'envs_per_gpu': n, 'global_envs': world * n, 'reset_pool': pool, 'reset': args.reset, 'parallelism': 'env-sharded x%d' % world,
                       'obs_allgather': bool(distributed), 'gather': gather_how, 'gathered_record': 'obs | reward | done | total_force_on_human | task_success' if distributed else None,
                       'gathered_in_global_order': in_order, 'noop_retest': blob.param('NOOP_RETEST'),
                       **({'ranks_share_gpus': '%d ranks on %d device(s): a rehearsal of the N-rank command (--share-gpus), not a scaling point' % (world, torch.cuda.device_count())} if getattr(args, 'share_gpus', False) else {})},
            'contacts_per_substep': contacts,      # solver contacts of the last substep of a step, mean over environments and sampled steps
            'overflow_count': int(overflow),       # substeps (summed over environments) in which a contact was dropped by the contact / row / coefficient budgets
            'pool_states_refreshed': int(getattr(env, 'pool_refreshed', 0)),      # --pool-refresh: start states a child process sampled during the run and the rollout swapped into the pool
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': names[dom], 'kernel_ms_per_launch': launch_ms, 'launches_per_step': launches[dom],
                         'chunks': chunks, 'envs_per_launch': envs_per_launch,
                         'algorithmic_bytes_per_env_step': bytes_per_env_step, 'record_bytes_per_env_step': record_bytes_per_env_step, 'algorithmic_bytes_per_launch': bytes_per_env_step * units,
                         'traffic_over_algorithmic': (traffic / (bytes_per_env_step * units)) if traffic else None,
                         'valu_issue_frac': valu_frac,      # of ONE launch (one chunk of environments); `chunks` such launches share the GPU
                         'valu_issue_frac_all_chunks': (valu_frac * chunks) if valu_frac else None,
                         'kernels_ms_per_step_summed_over_overlapping_launches': dict(zip(names, [float(x) for x in kms])),
                         'stream_ms_per_step': kernel_ms / K,
                         'step_level_achieved': bytes_per_env_step * n / (elapsed / K) / 1e9,
                         'note': 'solver bound by the latency of the dependent chain through the rows that share a body (wide row-local sweep: 1.38 rows per 38-instruction step, 56 % of the wave cycles at s_waitcnt: DESIGN 5), not by HBM; HBM fraction reported as the contract requires (SURVEY 8d)'},
        }
        if cloth_kernel:
            out['roofline']['cloth_kernel'] = cloth_kernel
        if world == 1 and cpu:
            out['cpu_baseline'] = cpu_baseline(blob, env.pool_host if args.reset != 'device' else env.stepper.get_state(), 8 if (not has_cloth or task == 'drinking') else 2, 1000 if not has_cloth else (300 if task == 'drinking' else 40),
                                               model + ('+coop' if blob.is_coop else ''), env_id,
                                               cloth=(env.cloth_pool_host if args.reset != 'device' else env.stepper.get_cloth()) if has_cloth else None)
    env.close()
    return out


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == '--cpu-worker':
        return _cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), *(sys.argv[5:6]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--pool', type=int, default=None, help='reset pool size (default 256; 64 for dressing, whose pool entries carry a garment and a 50-step device settle)')
    ap.add_argument('--reset', choices=['pool', 'device', 'host'], default='pool',
                    help="'pool': auto-reset from a fixed device-generated pool (BASELINE config 2); 'device': new states sampled for every episode")
    ap.add_argument('--pool-refresh', type=int, default=0, help='k > 0: a child process samples k new start states at a time and the rollout swaps them into the pool at episode boundaries (vec_env.PoolRefresher); the line then reports pool_states_refreshed')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--task', choices=sorted(TASKS), default=None, help="'feeding' = the BASELINE metric (config 2, the default); 'bedbathing' = config 3; 'scratchitch' = config 4's env (co-op) on one GPU; 'dressing' = config 5's")
    ap.add_argument('--workload', choices=['wiping', 'dense'], default=None, help="bedbathing only: 'wiping' = pad pressed onto the arm at every reset, 8-step episodes, small random actions; 'dense' = whole 200-step episodes under a scripted press-and-wipe policy (BASELINE config 3: dense tool-skin contact)")
    ap.add_argument('--env', default=None, help="any built env id instead of --task, e.g. 'ScratchItchJaco-v1' or 'FeedingSawyerHuman-v1' (assistive_gym_amd.envs.ENV_IDS)")
    ap.add_argument('--param', action='append', default=[], help='override a PARAMS entry of the model blob, e.g. --param NOOP_RETEST=0 (same-box A/B runs)')
    ap.add_argument('--no-configs', action='store_true', help='the default 1-GPU run also times short runs of BASELINE configs 3, 4 (1 GPU), 5 (1 GPU) into "configs"; this skips them')
    ap.add_argument('--backend', default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --dry-run)")
    ap.add_argument('--force-gather', action='store_true', help='1 GPU: run the multi-GPU code path anyway (a 1-rank RCCL process group, the per-step observation all-gather on its side stream) -- what the stream / hardware-queue layout of --gpus N looks like on one GPU')
    ap.add_argument('--gather', choices=['abi', 'torch'], default='abi', help="the collective of the per-step whole-batch gather with --gpus N: 'abi' = agx_comm_init_rank / agx_allgather of the C ABI (RCCL bound by libagx), 'torch' = torch.distributed")
    ap.add_argument('--share-gpus', action='store_true', help='REHEARSAL of the N-rank command on a box with fewer GPUs than ranks: rank r steps on device r %% device_count (several ranks share a GPU; RCCL refuses two ranks on one device, so use --backend gloo: the C-ABI collective then fails its vote and the gather falls back to torch.distributed, loudly).  The line says so; its value is not a scaling point')
    ap.add_argument('--dry-run', action='store_true', help='exercise the launch / sharding / all-gather / timing path on CPU tensors without stepping (no GPU needed; backend gloo)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1) -- the same command
        # line the driver uses for N > 1; the ranks see WORLD_SIZE and take the branch below
        import socket
        import subprocess
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        raise SystemExit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
                                          '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    if args.dry_run:
        import torch.distributed as dist
        world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
        if world > 1:
            dist.init_process_group(args.backend or 'gloo')
        out = dry_run(args, rank, world)
        if args.task is None and args.env is None and world in MULTI_GPU_CONFIGS and not args.no_configs:
            key, t, _ = MULTI_GPU_CONFIGS[world]
            r = dry_run(argparse.Namespace(**dict(vars(args), task=t)), rank, world)
            out.setdefault('configs', {})[key] = dict({k: r[k] for k in ('value', 'unit', 'n_gpus', 'steps', 'ms_per_step')}, workload=r['config']['workload'], global_envs=r['config']['global_envs'],
                                                       gather=r['config']['gather'], gathered_in_global_order=r['config']['gathered_in_global_order'])
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return
    args.backend = args.backend or 'nccl'

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1 or args.force_gather
    if args.force_gather and world == 1:
        import socket
        sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
        for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', str(port)), ('RANK', '0'), ('WORLD_SIZE', '1'), ('LOCAL_RANK', '0')):
            os.environ.setdefault(k, v)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: libagx has no CPU path')
    if args.share_gpus:
        local_rank %= torch.cuda.device_count()       # (local_rank is only ever used as the device index below)
    torch.cuda.set_device(local_rank)
    if distributed:
        if args.backend == 'nccl':
            dist.init_process_group(args.backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    headline = args.task is None and args.env is None
    task = args.task or 'feeding'
    out = run_config(args, task, args.steps, args.warmup, rank, world, local_rank, distributed, cpu=not args.no_cpu_baseline, workload=args.workload, env_id_override=args.env)
    if headline and world == 1 and not args.no_configs:
        # beside the headline: the same workload over 2,000 steps (ten episode boundaries inside the window) when the timed
        # window was short, and with the EXACT solver -- NOOP_RETEST = 0: every row visited in every sweep, no re-test rule -- over 300 steps
        if args.steps < 200:
            r = run_config(args, task, 2000, 50, rank, world, local_rank, distributed, cpu=False)
            out['value_2000_steps'] = r['value']
        args0 = argparse.Namespace(**dict(vars(args), param=list(args.param) + ['NOOP_RETEST=0']))
        r = run_config(args0, task, 300, 20, rank, world, local_rank, distributed, cpu=False)
        out['value_noop_retest_0'] = r['value']
        out['value_noop_retest_0_note'] = 'same process, 300 steps, --param NOOP_RETEST=0: plain projected Gauss-Seidel, every row visited in all %d sweeps' % int(r['config']['workload'].split(',')[-1].split()[0])
    if headline and world in MULTI_GPU_CONFIGS and not args.no_configs:
        # the driver's 4- and 8-rank commands also run BASELINE config 4 (ScratchItchPR2 co-op, 16,384 environments over 4 GPUs) / config 5
        # (DressingBaxter, 32,768 over 8) as written: same sharding, same per-step whole-batch gather (every rank takes part: collectives)
        key, t, st = MULTI_GPU_CONFIGS[world]
        r = run_config(args, t, st, 10, rank, world, local_rank, distributed, cpu=False)
        if rank == 0:
            out.setdefault('configs', {})[key] = dict({k: r[k] for k in ('value', 'unit', 'n_gpus', 'steps', 'ms_per_step', 'contacts_per_substep', 'overflow_count')},
                                                       workload=r['config']['workload'], global_envs=r['config']['global_envs'], gather=r['config']['gather'],
                                                       gathered_in_global_order=r['config']['gathered_in_global_order'])
    if headline and world == 1 and not args.no_configs:
        # the other single-GPU BASELINE configurations on the same clock: short runs (a few seconds each), the headline value above is config 2
        extra = {}
        for key, t, wl, st in (('config3_BedBathingSawyer-v1', 'bedbathing', None, 300), ('config3_BedBathingSawyer-v1_wiping_contact', 'bedbathing', 'wiping', 300),
                               ('config3_BedBathingSawyer-v1_dense', 'bedbathing', 'dense', 400),
                               ('config4_ScratchItchPR2Human-v1_1gpu', 'scratchitch', None, 300), ('config5_DressingBaxter-v1_1gpu', 'dressing', None, 30)):
            r = run_config(args, t, st, 10, rank, world, local_rank, distributed, cpu=False, workload=wl)
            extra[key] = {k: r[k] for k in ('value', 'unit', 'steps', 'ms_per_step', 'contacts_per_substep', 'overflow_count')}
            extra[key]['workload'] = r['config']['workload']
            extra[key]['roofline'] = {k: r['roofline'][k] for k in ('achieved', 'peak', 'frac', 'traffic', 'kernel', 'kernel_ms_per_launch', 'algorithmic_bytes_per_env_step', 'valu_issue_frac') if k in r['roofline']}
            if 'cloth_kernel' in r['roofline']:
                extra[key]['roofline']['cloth_kernel'] = r['roofline']['cloth_kernel']
        # config 3 with a NEW human / base pose for every episode of every environment (reset='device': 4096 rag dolls dropped and settled for
        # 100 steps at each 200-step boundary -- the Amdahl term the pool avoids; two boundaries inside the timed window) beside the pool rate above
        args_dev = argparse.Namespace(**dict(vars(args), reset='device', pool=None))
        r = run_config(args_dev, 'bedbathing', 400, 10, rank, world, local_rank, distributed, cpu=False)
        extra['config3_BedBathingSawyer-v1_reset_on_device_every_episode'] = {k: r[k] for k in ('value', 'unit', 'steps', 'ms_per_step', 'contacts_per_substep', 'overflow_count')}
        extra['config3_BedBathingSawyer-v1_reset_on_device_every_episode']['workload'] = r['config']['workload'] + '; every episode starts from a newly sampled and settled state (agx_reset with the rag-doll model attached)'
        # Drinking: the sixth task, 64 water particles per environment
        r = run_config(args, 'drinking', 200, 10, rank, world, local_rank, distributed, cpu=False)
        extra['DrinkingJaco-v1'] = {k: r[k] for k in ('value', 'unit', 'steps', 'ms_per_step', 'contacts_per_substep', 'overflow_count')}
        extra['DrinkingJaco-v1']['workload'] = r['config']['workload']
        out['configs'] = extra
    if distributed:
        # RCCL writes a version banner through C stdio when a communicator is created; buffered, it would come out at exit -- BEHIND the JSON line.
        # The one JSON line is the last thing this process prints: flush C stdio first, print, and tear the process group down quietly.
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
