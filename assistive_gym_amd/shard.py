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


class ObsGatherer:
    """Whole-batch observation collation overlapped with the next env.step().

    The all-gather of step k runs on a SIDE stream: it waits (event) for the kernels that produced the observations of step k
    and then proceeds concurrently with the kernels of step k + 1 on the compute stream.  Two observation buffers and two
    gathered buffers alternate, so step k + 1 never writes what the gather of step k is still reading; before a buffer is
    reused (step k + 2) the compute stream waits for the gather that read it.  `wait(slot)` makes the current stream wait for
    the gathered batch of that slot (what a learner consuming the whole batch calls).
    On CPU tensors (gloo tests) there are no streams: submit() gathers synchronously.
    """

    def __init__(self, n_local, obs_dim, world, device=None, dtype=None, force=False):
        import torch
        self.torch, self.world = torch, world
        self.force = force            # a 1-rank group still runs the collective (bench.py --force-gather: the stream layout of N ranks on one GPU)
        self.cuda = device is not None and torch.device(device).type == 'cuda'
        dtype = dtype or torch.float32
        self.local = [torch.zeros((n_local, obs_dim), dtype=dtype, device=device) for _ in range(2)]
        self.full = [torch.zeros((world * n_local, obs_dim), dtype=dtype, device=device) for _ in range(2)]
        if self.cuda:
            self.stream = torch.cuda.Stream(device=device)
            self.produced = [torch.cuda.Event() for _ in range(2)]
            self.gathered = [torch.cuda.Event() for _ in range(2)]
            self.pending = [False, False]

    def buffer(self, slot):
        """the observation buffer step `slot` (= step index mod 2) writes into; on the GPU the compute stream first waits for
        the gather that last read this buffer"""
        if self.cuda and self.pending[slot]:
            self.torch.cuda.current_stream().wait_event(self.gathered[slot])
            self.pending[slot] = False
        return self.local[slot]

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
