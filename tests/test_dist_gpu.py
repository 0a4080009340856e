"""-m gpu: the sharded path on real hardware with whatever GPUs are visible.  Two ranks step their shards (on the same GPU when
only one is visible -- RCCL refuses two ranks on one device, so the collation of this test goes through gloo on the host) and the
per-environment results are compared BIT FOR BIT with a one-rank run of the whole batch, across an episode boundary (auto-reset
from the pool by global env index).  The side-stream protocol of shard.ObsGatherer is exercised with a stand-in collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS, SEED = 204, 4242
# BASELINE config 2's env and the two multi-GPU configs (4: ScratchItchPR2 co-op, 17 actions, 30 + 34 observations; 5: DressingBaxter, whose
# pool entries carry a garment each): (VecEnv class, global envs, pool size)
CASES = {'feeding': ('FeedingJacoVecEnv', 48, 5), 'scratchitch_coop': ('ScratchItchPR2HumanVecEnv', 16, 3), 'dressing': ('DressingBaxterVecEnv', 8, 3)}


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _rollout(case, n, env_offset, device):
    from assistive_gym_amd import vec_env
    cls, n_global, pool = CASES[case]
    env = getattr(vec_env, cls)(n, device=device, seed=SEED, pool_size=pool)
    env.reset(env_offset=env_offset)
    tape = torch.from_numpy(np.random.RandomState(7).uniform(-1, 1, (STEPS, n_global, env.act_dim)).astype(np.float32))
    obs_log, rew_log = [], []
    for k in range(STEPS):
        obs, rew, done, info = env.step(tape[k, env_offset:env_offset + n].contiguous().cuda(device))
        if k % 17 == 0 or k >= STEPS - 5:
            obs_log.append(obs.cpu().clone()); rew_log.append(rew.cpu().clone())
    final = torch.from_numpy(env.stepper.get_state())
    if getattr(env, 'cloth_pool_host', None) is not None:          # the garments are part of the state: appended, one row per environment
        final = torch.cat([final, torch.from_numpy(env.stepper.get_cloth()).reshape(n, -1)], dim=1)
    env.close()
    return torch.stack(obs_log), torch.stack(rew_log), final


def _worker(rank, world, port, q, case):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = rank % torch.cuda.device_count()
    n = CASES[case][1] // world
    obs, rew, final = _rollout(case, n, rank * n, dev)
    outs = []
    for t in (obs.transpose(0, 1).contiguous(), rew.transpose(0, 1).contiguous(), final):      # env-major, so that shards concatenate
        full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(full, t)
        outs.append(full.numpy().copy())
    if rank == 0:
        q.put(outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('case', sorted(CASES))
def test_two_ranks_match_one_rank_bit_for_bit(case):
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    obs1, rew1, final1 = _rollout(case, CASES[case][1], 0, 0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, case)) for r in range(2)]
    [p.start() for p in ps]
    obs2, rew2, final2 = q.get(timeout=600)
    [p.join(timeout=120) for p in ps]
    assert np.array_equal(obs2, obs1.transpose(0, 1).numpy().view(np.float32)) or np.array_equal(obs2.view(np.int32), obs1.transpose(0, 1).contiguous().numpy().view(np.int32))
    assert np.array_equal(rew2.view(np.int32), rew1.transpose(0, 1).contiguous().numpy().view(np.int32))
    assert np.array_equal(final2.view(np.int32), final1.numpy().view(np.int32))       # incl. the states drawn from the pool after step 200


def test_gatherer_overlaps_on_a_side_stream(monkeypatch):
    """the double-buffer / event protocol on the GPU with a stand-in collective (two copies of the local shard)"""
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd.shard import ObsGatherer
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    calls = []

    def fake_all_gather(out, inp):
        calls.append(torch.cuda.current_stream().cuda_stream)
        n = inp.shape[0]
        out[:n].copy_(inp); out[n:].copy_(inp)
    monkeypatch.setattr(dist, 'all_gather_into_tensor', fake_all_gather)
    n = 64
    env = FeedingJacoVecEnv(n, pool_size=4, seed=5)
    env.reset()
    g = ObsGatherer(n, env.obs_dim, 2, device=env.device)
    gen = torch.Generator(device='cuda'); gen.manual_seed(1)
    seen = []
    for k in range(6):
        a = torch.rand((n, 7), device='cuda', generator=gen) * 2 - 1
        obs, _, _, _ = env.step(a, obs_out=g.buffer(k & 1))
        full = g.submit(k & 1)
        g.wait(k & 1)
        torch.cuda.synchronize()
        assert torch.equal(full[:n], obs) and torch.equal(full[n:], obs)
        seen.append(obs.clone())
    assert not torch.equal(seen[-1], seen[-2])
    assert all(s == g.stream.cuda_stream for s in calls) and g.stream.cuda_stream != torch.cuda.current_stream().cuda_stream
    env.close()


def test_c_abi_allgather_single_rank():
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.libagx import Stepper
    st = Stepper(ModelBlob.load('feeding_jaco'), 4)
    a = torch.arange(100, dtype=torch.float32, device='cuda'); b = torch.zeros(100, device='cuda')
    st.allgather(a, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    st.close()


def test_c_abi_packed_gather_through_a_communicator():
    """The whole-batch record of a step packed on the device (agx_pack_step) and gathered by the C ABI's own collective (agx_comm_unique_id /
    agx_comm_init_rank / agx_allgather: RCCL bound by libagx) on the gatherer's side stream -- a one-rank communicator here (RCCL takes one rank
    per device; the N-rank run is bench.py --gpus N, whose default is this path) -- against the same record packed with tensor ops."""
    if not torch.cuda.is_available():
        __import__('conftest').no_gpu()
    from assistive_gym_amd import libagx
    from assistive_gym_amd.shard import BatchGatherer
    from assistive_gym_amd.vec_env import FeedingJacoVecEnv
    n = 64
    env = FeedingJacoVecEnv(n, pool_size=8, seed=5)
    env.reset()
    comm = libagx.comm_init_rank(0, 0, 1, libagx.comm_unique_id())
    assert comm
    g = BatchGatherer(n, env.obs_dim + 4, 1, device=torch.device('cuda', 0), force=True, stepper=env.stepper, comm=comm)
    gen = torch.Generator(device='cuda'); gen.manual_seed(3)
    for k in range(4):
        obs, rew, done, info = env.step(torch.rand((n, env.act_dim), device='cuda', generator=gen) * 2 - 1)
        g.pack(k & 1, obs, rew, done, info)
        full = g.submit(k & 1)
        g.wait(k & 1)
        torch.cuda.synchronize()
        want = torch.cat([obs, rew[:, None], done.float()[:, None], info[:, 0:2]], dim=1)
        assert full.shape == (n, env.obs_dim + 4) and torch.equal(full, want)
    libagx.comm_destroy(comm)
    env.close()
