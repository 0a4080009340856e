"""Environment sharding across GPUs: independent environments, contiguous index ranges
per rank, no data-path collective; one all-gather of the observation shards only when a single
consumer wants the whole batch (RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
import numpy as np


def shard_range(rank, world, global_envs):
    """Contiguous [lo, hi) range of global env indices owned by `rank`."""
    per = global_envs // world
    assert per * world == global_envs, 'global env count must divide evenly'
    return rank * per, (rank + 1) * per


def pool_indices(env_offset, n, pool_size):
    """Initial-state pool entry of each local env: a function of the GLOBAL env index only, so an
    environment's trajectory does not depend on which GPU it lands on."""
    return (np.arange(n) + env_offset) % pool_size


def episode_seed(seed, episode, env_offset):
    """Seed handed to the device-side reset generator for the first env of a shard: env i of episode e is sampled from
    seed + e * 2^32 + (env_offset + i), a function of the GLOBAL env index and the episode only."""
    return (int(seed) + (int(episode) << 32) + int(env_offset)) & 0xFFFFFFFFFFFFFFFF


def gather_observations(obs_local, world, out=None):
    """[n, obs_dim] per rank -> [world*n, obs_dim] in rank order on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return obs_local
    if out is None:
        out = torch.empty((world * obs_local.shape[0], obs_local.shape[1]), dtype=obs_local.dtype, device=obs_local.device)
    dist.all_gather_into_tensor(out, obs_local.contiguous())
    return out


class BatchGatherer:
    """Whole-batch collation overlapped with the next env.step().  What is gathered per step is ONE record per environment:
    observation | reward | done | total_force_on_human | task_success ([n, obs_dim + 4] float32, `pack()`; SURVEY 8e: what the single consumer
    of the whole batch -- the sampler of learn.py:26,72 -- reads), or any [n, width] buffer the caller fills.

    The all-gather of step k runs on a SIDE stream: it waits (event) for the kernels that produced the records of step k and then proceeds
    concurrently with the kernels of step k + 1 on the compute stream.  Two local and two gathered buffers alternate, so step k + 1 never
    writes what the gather of step k is still reading; before a buffer is reused (step k + 2) the compute stream waits for the gather that
    read it.  `wait(slot)` makes the current stream wait for the gathered batch of that slot (what a learner consuming it calls).
    The collective: torch.distributed (backend nccl = RCCL; gloo on CPU tensors in the tests, synchronously -- no streams there), or
    `comm` = a communicator of the C ABI (libagx.comm_init_rank) together with `stepper`: agx_allgather, RCCL bound by libagx itself.
    """

    def __init__(self, n_local, width, world, device=None, dtype=None, force=False, stepper=None, comm=None):
        import torch
        self.torch, self.world = torch, world
        self.force = force            # a 1-rank group still runs the collective (bench.py --force-gather: the stream layout of N ranks on one GPU)
        self.cuda = device is not None and torch.device(device).type == 'cuda'
        self.stepper, self.comm = stepper, comm
        assert comm is None or (stepper is not None and self.cuda), 'the C ABI collective gathers device buffers of a stepper handle'
        dtype = dtype or torch.float32
        self.local = [torch.zeros((n_local, width), dtype=dtype, device=device) for _ in range(2)]
        self.full = [torch.zeros((world * n_local, width), dtype=dtype, device=device) for _ in range(2)]
        if self.cuda:
            self.stream = torch.cuda.Stream(device=device)
            self.produced = [torch.cuda.Event() for _ in range(2)]
            self.gathered = [torch.cuda.Event() for _ in range(2)]
            self.pending = [False, False]

    def buffer(self, slot):
        """the local buffer step `slot` (= step index mod 2) writes into; on the GPU the compute stream first waits for the gather that
        last read this buffer"""
        if self.cuda and self.pending[slot]:
            self.torch.cuda.current_stream().wait_event(self.gathered[slot])
            self.pending[slot] = False
        return self.local[slot]

    def pack(self, slot, obs, reward, done, info):
        """local[slot] = observation | reward | done | info[:, 0:2] of this step, on the current stream (agx_pack_step on the device; the same
        layout with tensor ops on CPU tensors)"""
        out = self.buffer(slot)
        od = obs.shape[1]
        assert out.shape[1] == od + 4
        if self.cuda and self.stepper is not None:
            self.stepper.pack_step(obs, reward, done, info, out, self.torch.cuda.current_stream().cuda_stream)
        else:
            out[:, :od] = obs; out[:, od] = reward; out[:, od + 1] = done.to(out.dtype); out[:, od + 2:od + 4] = info[:, 0:2]
        return out

    def submit(self, slot):
        """enqueue the all-gather of local[slot] -> full[slot]"""
        import torch.distributed as dist
        if self.world == 1 and not self.force:
            self.full[slot] = self.local[slot]
            return self.full[slot]
        if not self.cuda:
            dist.all_gather_into_tensor(self.full[slot], self.local[slot].contiguous())
            return self.full[slot]
        torch = self.torch
        self.produced[slot].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.produced[slot])
            if self.comm is not None:
                self.stepper.allgather(self.local[slot], self.full[slot], self.comm, self.stream.cuda_stream)
            else:
                dist.all_gather_into_tensor(self.full[slot], self.local[slot])
            self.gathered[slot].record(self.stream)
        self.pending[slot] = True
        return self.full[slot]

    def wait(self, slot=None):
        """the current stream waits for the gather of `slot` (both when None)"""
        if not self.cuda:
            return
        for s in ((0, 1) if slot is None else (slot,)):
            if self.pending[s]:
                self.torch.cuda.current_stream().wait_event(self.gathered[s])
                self.pending[s] = False


ObsGatherer = BatchGatherer      # (rounds 2-4: observations only)
