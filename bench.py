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

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SHADER_CLOCK_HZ = 2.4e9   # same guide: max clock 2400 MHz; 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles
N_SIMD = 1024
TASKS = {   # --task -> (model blob, VecEnv class, kernel-name suffix of the compiled variant, workload description)
    'feeding': ('feeding_jaco', 'FeedingJacoVecEnv', '', 'FeedingJaco-v1'),
    'feedingpanda': ('feeding_panda', 'FeedingPandaVecEnv', '', 'FeedingPanda-v1'),
    'bedbathing': ('bed_bathing_sawyer', 'BedBathingSawyerVecEnv', '_bb', 'BedBathingSawyer-v1'),
    'scratchitch': ('scratch_itch_pr2', 'ScratchItchPR2HumanVecEnv', '_si', 'ScratchItchPR2Human-v1 (co-op: 7 robot + 10 human actions)'),
    'armmanipulation': ('arm_manipulation_sawyer', 'ArmManipulationSawyerVecEnv', '_am', 'ArmManipulationSawyer-v1 (14 actions: the single arm listed twice)'),
    'feedingsawyer': ('feeding_sawyer', 'FeedingSawyerVecEnv', '_fl', 'FeedingSawyer-v1 (free-standing robot: feeding_l kernel variant, 320 colliders)'),
    'bedbathingpr2': ('bed_bathing_pr2', 'BedBathingPR2VecEnv', '_bbl', 'BedBathingPR2-v1 (bed_bathing_l kernel variant: 24 DoF)'),
    'scratchitchjaco': ('scratch_itch_jaco', 'ScratchItchJacoVecEnv', '_si', "ScratchItchJaco-v1 (the reference's default environment)"),
    'dressing': ('dressing_baxter', 'DressingBaxterVecEnv', '_dr', 'DressingBaxter-v1 (cloth of 3,966 nodes per env, numSubSteps 8: 40 internal substeps per step)'),
}


def _cpu_worker(path, seed, n_steps, model='feeding_jaco'):
    """`bench.py --cpu-worker`: one host process stepping its share of the sample with the C oracle;
    prints "<env-steps> <seconds>"."""
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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--task', choices=sorted(TASKS), default='feeding', help="'feeding' = the BASELINE metric (config 2); 'bedbathing' = config 3; 'scratchitch' = config 4's env (co-op) on one GPU")
    ap.add_argument('--env', default=None, help="any built env id instead of --task, e.g. 'ScratchItchJaco-v1' or 'FeedingSawyerHuman-v1' (assistive_gym_amd.envs.ENV_IDS)")
    args = ap.parse_args()
    model, env_cls, ksuffix, env_id = TASKS[args.task]
    if args.env is not None:             # the kernel-name suffix is that of the variant agx_create picks: taken from the stepper below
        from assistive_gym_amd.envs import ENV_IDS
        cls = ENV_IDS[args.env.split(':')[-1]]
        model, env_cls, ksuffix, env_id = cls.model, None, None, args.env.split(':')[-1]
        args.task = 'dressing' if model.startswith('dressing') else args.env
    if args.pool is None:
        args.pool = 64 if args.task == 'dressing' else 256

    import torch
    import torch.distributed as dist
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd import vec_env
    from assistive_gym_amd.shard import ObsGatherer

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: libagx has no CPU path')
    torch.cuda.set_device(local_rank)
    if distributed:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    n = args.envs_per_gpu
    if env_cls is None:
        env = vec_env.AssistiveVecEnv(n, device=local_rank, seed=1001, pool_size=args.pool, reset=args.reset, model=model, coop=env_id.endswith('Human-v1'))
        ksuffix = {'feeding': '', 'feeding_l': '_fl', 'bed_bathing': '_bb', 'bed_bathing_l': '_bbl', 'scratch_itch': '_si', 'dressing': '_dr', 'dressing_l': '_drl',
                   'arm_manipulation': '_am', 'arm_manipulation_l': '_aml'}[env.stepper.variant()]
    else:
        env = getattr(vec_env, env_cls)(n, device=local_rank, seed=1001, pool_size=args.pool, reset=args.reset)
    blob = env.blob                      # the co-op flavour where the task's BASELINE config is co-op
    env.reset(env_offset=rank * n)
    K, W = args.steps, args.warmup
    g = torch.Generator(device='cuda'); g.manual_seed(1001 + rank)
    tape = torch.rand((W + K, n, blob.act_dim), device='cuda', generator=g) * 2 - 1
    # whole-batch observation collation: RCCL all-gather on a side stream, overlapped with the next step (two buffers alternate)
    gatherer = ObsGatherer(n, blob.obs_dim, world, device=torch.device('cuda', local_rank)) if distributed else None

    def one(k):
        if distributed:
            env.step(tape[k], obs_out=gatherer.buffer(k & 1))
            gatherer.submit(k & 1)
        else:
            env.step(tape[k])

    for k in range(W):
        one(k)
    if distributed:
        gatherer.wait()
    stream = torch.cuda.current_stream().cuda_stream
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    env.stepper.profile_begin(stream)
    t0 = time.perf_counter()
    for k in range(W, W + K):
        one(k)
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
    # per-kernel launch durations (HIP events after every launch, on the chunk stream it is launched on),
    # measured after the timed region so that `value` is not perturbed by the extra events / host syncs.
    # The step is issued as a few independent chunks of environments on internal streams, so launches of
    # different chunks overlap: the durations are those of launches that share the GPU, exactly what a
    # kernel trace (rocprofv3 --kernel-trace) of this command reports.
    NT = 20
    kms, kcnt = np.zeros(3), np.zeros(3)
    has_cloth = getattr(env, 'cloth_pool_host', None) is not None
    if has_cloth:
        # 40 build / solve pairs + the cloth kernel + finish per step: too many launches for the per-launch event table; the split by
        # kernel comes from the rocprofv3 kernel trace (profiles/); here the whole step is the unit
        kms[:] = [0, 0, kernel_ms / K]; kcnt[:] = [0, 0, env.stepper.n_chunks() if hasattr(env.stepper, 'n_chunks') else 1]
    else:
        for k in range(NT):
            ms, cnt = env.stepper.step_timed(tape[(W + k) % (W + K)], env.obs, env.reward, env.done, env.info, stream)
            kms += np.array(ms); kcnt += np.array(cnt)
        kms /= NT; kcnt /= NT
    if rank == 0:
        total_steps = world * n * K
        value = total_steps / elapsed
        sw = blob.state_words
        fs = int(blob.param('FRAME_SKIP'))
        # algorithmic HBM bytes per env-step: state record read + written once, action read, obs /
        # reward / done / info written (DESIGN.md "bytes per env-step")
        bytes_per_env_step = 2 * sw * 4 + blob.act_dim * 4 + blob.obs_dim * 4 + 4 + 1 + 8 * 4
        if has_cloth:                        # + the garment read and written once per env step: node positions and velocities
            bytes_per_env_step += 2 * env.cloth_pool_host[0].size * 4
        names = [k + ksuffix for k in ('agx_build_kernel', 'agx_solve_kernel', 'agx_finish_kernel')]
        if has_cloth:
            names[2] = 'whole step (40 x [agx_build_kernel%s, agx_solve_kernel%s] + agx_cloth_kernel%s + agx_finish_kernel%s)' % ((ksuffix,) * 4)
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
        traffic, traffic_src, valu_frac = None, None, None
        for cand in ('r02_traffic_%s.json' % args.task, 'r01_traffic.json' if args.task == 'feeding' else None):
            tpath = cand and os.path.join(ROOT, 'profiles', cand)
            if tpath and os.path.exists(tpath):
                tj = json.load(open(tpath))
                kj = tj.get('kernels', {}).get(names[dom])
                if kj:
                    traffic = kj['hbm_bytes_per_env_launch'] * envs_per_launch
                    traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s), per environment x %d environments per launch' % (cand, tj.get('correction', ''), envs_per_launch)
                    if 'valu_insts_per_env_launch' in kj:
                        # wave64 VALU instruction = 2 issue cycles on a SIMD-32 (guide, "Wave scheduling"); 1024 SIMDs
                        valu_frac = kj['valu_insts_per_env_launch'] * envs_per_launch * 2.0 / (launch_ms * 1e-3 * SHADER_CLOCK_HZ * N_SIMD)
                    break
        out = {
            'metric': 'env_steps_per_sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s, %d lockstep envs per MI355X, random-policy rollout, 5 simulation steps per env step, 50 PGS sweeps' % (env_id, n),
                       'envs_per_gpu': n, 'global_envs': world * n, 'reset_pool': args.pool, 'reset': args.reset, 'parallelism': 'env-sharded x%d' % world,
                       'obs_allgather': bool(distributed)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': names[dom], 'kernel_ms_per_launch': launch_ms, 'launches_per_step': launches[dom],
                         'chunks': chunks, 'envs_per_launch': envs_per_launch,
                         'algorithmic_bytes_per_env_step': bytes_per_env_step, 'algorithmic_bytes_per_launch': bytes_per_env_step * units,
                         'traffic_over_algorithmic': (traffic / (bytes_per_env_step * units)) if traffic else None,
                         'valu_issue_frac': valu_frac,      # of ONE launch (one chunk of environments); `chunks` such launches share the GPU
                         'valu_issue_frac_all_chunks': (valu_frac * chunks) if valu_frac else None,
                         'kernels_ms_per_step_summed_over_overlapping_launches': dict(zip(names, [float(x) for x in kms])),
                         'stream_ms_per_step': kernel_ms / K,
                         'step_level_achieved': bytes_per_env_step * n / (elapsed / K) / 1e9,
                         'note': 'dependent-chain latency bound solver (VALU issue ~0.3 of peak); HBM fraction reported as the contract requires (SURVEY 8d)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(blob, env.pool_host if args.reset != 'device' else env.stepper.get_state(), 8 if not has_cloth else 2, 1000 if not has_cloth else 40,
                                               model + ('+coop' if blob.is_coop else ''), env_id, cloth=env.cloth_pool_host if has_cloth else None)
        print(json.dumps(out))
    env.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
