"""Batched FeedingJaco-v1 environments on one GPU (torch tensors in / out, state resident in HBM).

The scalar gym.Env facade with the reference's class names lives in assistive_gym_amd/envs.py;
this is the data-parallel form: N lock-stepped copies of FeedingJacoEnv
(assistive_gym/envs/feeding_envs.py:29-31), stepped by one kernel launch per env.step().
"""
import numpy as np
import torch

from .blob import ModelBlob
from .host.reset import make_states
from .libagx import Stepper
from .shard import pool_indices

SETTLE_STEPS = 25   # feeding.py:178-179


def build_reset_pool(blob, pool_size, seed, device=0, impairment='random'):
    """pool_size post-reset states: host-side sampling + IK (host/reset.py), then the 25 settle
    steps of feeding.py:178-179 on the device.  Returns a float32 (pool_size, state_words) array."""
    states, _ = make_states(blob, pool_size, seed=seed, impairment=impairment)
    st = Stepper(blob, pool_size, device)
    st.set_state(states)
    st.settle(SETTLE_STEPS)
    st.synchronize()
    out = st.get_state()
    st.close()
    return out


class FeedingJacoVecEnv:
    def __init__(self, n_envs, device=0, seed=1001, pool_size=256, blob=None, impairment='random', auto_reset=True):
        self.blob = blob or ModelBlob.load('feeding_jaco')
        self.n_envs, self.device_index, self.seed = n_envs, device, seed
        self.device = torch.device('cuda', device)
        self.pool_size, self.impairment, self.auto_reset = pool_size, impairment, auto_reset
        self.stepper = Stepper(self.blob, n_envs, device)
        self.act_dim, self.obs_dim = self.blob.act_dim, self.blob.obs_dim
        self.obs = torch.zeros((n_envs, self.obs_dim), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(n_envs, dtype=torch.float32, device=self.device)
        self.done = torch.zeros(n_envs, dtype=torch.uint8, device=self.device)
        self.info = torch.zeros((n_envs, 8), dtype=torch.float32, device=self.device)
        self.pool = None

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, env_offset=0):
        """env_offset: global index of this shard's first env (multi-GPU sharding keeps the
        env -> initial state mapping independent of the GPU count)."""
        if self.pool is None:
            self.pool_host = build_reset_pool(self.blob, self.pool_size, self.seed, self.device_index, self.impairment)
            self.pool = torch.from_numpy(self.pool_host).to(self.device)
        idx = pool_indices(env_offset, self.n_envs, self.pool_size)
        self.stepper.set_state(self.pool_host[idx])
        self.stepper.observe_dev(self.obs, self._stream())
        return self.obs

    def step(self, actions):
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.shape == (self.n_envs, self.act_dim) and actions.is_contiguous()
        s = self._stream()
        self.stepper.step_dev(actions, self.obs, self.reward, self.done, self.info, s)
        if self.auto_reset:
            self.stepper.reset_done(self.pool, self.pool_size, self.done, s)
        return self.obs, self.reward, self.done, self.info

    def close(self):
        self.stepper.close()
