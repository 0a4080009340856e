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
