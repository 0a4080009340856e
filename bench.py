#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched FeedingJaco-v1 stepper (BASELINE.json metric).

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


def _cpu_worker(path, seed, n_steps):
    """`bench.py --cpu-worker`: one host process stepping its share of the sample with the C oracle;
    prints "<env-steps> <seconds>"."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_lib import Oracle
    from assistive_gym_amd.blob import ModelBlob
    blob = ModelBlob.load('feeding_jaco')
    o = Oracle(blob)
    init = np.load(path)
    st = init.copy()
    rng = np.random.RandomState(seed)
    t0 = time.perf_counter()
    for k in range(n_steps):
        if k and k % 200 == 0:
            st[:] = init            # episode end: back to a post-reset state, like the device-side auto-reset
        a = rng.uniform(-1, 1, (len(st), blob.act_dim)).astype(np.float32)
        for i in range(len(st)):
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


def cpu_baseline(blob, states, envs_per_core, n_steps):
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
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', path, str(c), str(n_steps)],
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
                sample='%d processes x %d envs x %d steps of the same FeedingJaco workload, C f64 oracle (not PyBullet); '
                       'per process %.0f env-steps/s; wall incl. start-up %.1f s' % (cores, envs_per_core, n_steps, single, wall))


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == '--cpu-worker':
        return _cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--pool', type=int, default=256)
    ap.add_argument('--reset', choices=['pool', 'device', 'host'], default='pool',
                    help="'pool': auto-reset from a fixed device-generated pool (BASELINE config 2); 'device': new states sampled for every episode")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    from assistive_gym_amd.shard import gather_observations

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
    blob = ModelBlob.load('feeding_jaco')
    env = FeedingJacoVecEnv(n, device=local_rank, seed=1001, pool_size=args.pool, reset=args.reset)
    env.reset(env_offset=rank * n)
    K, W = args.steps, args.warmup
    g = torch.Generator(device='cuda'); g.manual_seed(1001 + rank)
    tape = torch.rand((W + K, n, blob.act_dim), device='cuda', generator=g) * 2 - 1
    gathered = torch.empty((world * n, blob.obs_dim), device='cuda') if distributed else None

    def one(k):
        obs, rew, done, info = env.step(tape[k])
        if distributed:
            gather_observations(obs, world, gathered)

    for k in range(W):
        one(k)
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
        names = ['agx_build_kernel', 'agx_solve_kernel', 'agx_finish_kernel']
        dom = int(np.argmax(kms))
        # one launch of the dominant kernel advances the environments of one chunk by 1/frame_skip of an env-step
        launches = [int(round(x)) for x in kcnt]
        launch_ms = kms[dom] / launches[dom]
        units = n / launches[dom]            # env-steps advanced by one launch (all launches of a kind together: n)
        achieved = bytes_per_env_step * units / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
        if os.path.exists(tpath):       # PMC passes of tools/pmc_workload.py (separate rocprofv3 runs)
            tj = json.load(open(tpath))
            if tj.get('envs') == n and names[dom] in tj.get('kernels', {}):
                traffic = tj['kernels'][names[dom]]['hbm_bytes_per_launch']
                traffic_src = 'profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s)' % tj.get('correction', '')
        out = {
            'metric': 'env_steps_per_sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'FeedingJaco-v1, %d lockstep envs per MI355X, random-policy rollout, 5 substeps/step, 50 PGS sweeps' % n,
                       'envs_per_gpu': n, 'global_envs': world * n, 'reset_pool': args.pool, 'reset': args.reset, 'parallelism': 'env-sharded x%d' % world,
                       'obs_allgather': bool(distributed)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': names[dom], 'kernel_ms_per_launch': launch_ms, 'launches_per_step': launches[dom],
                         'chunks': launches[2], 'envs_per_launch': n / launches[2],
                         'algorithmic_bytes_per_env_step': bytes_per_env_step, 'algorithmic_bytes_per_launch': bytes_per_env_step * units,
                         'kernels_ms_per_step_summed_over_overlapping_launches': dict(zip(names, [float(x) for x in kms])),
                         'stream_ms_per_step': kernel_ms / K,
                         'step_level_achieved': bytes_per_env_step * n / (elapsed / K) / 1e9,
                         'note': 'latency/VALU-bound solver; HBM fraction reported as the contract requires (SURVEY 8d)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(blob, env.pool_host if args.reset != 'device' else env.stepper.get_state(), 8, 1000)
        print(json.dumps(out))
    env.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
