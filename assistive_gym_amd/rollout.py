"""On-device PPO rollouts for the batched environments.

The reference trains with RLlib PPO on ``cpu_count()`` scalar gym workers (assistive_gym/learn.py:9-37:
train_batch_size 19200, lambda 0.95, fcnet_hiddens [100, 100]); observations and actions cross the
process boundary every step.  Here the policy (the same 2 x 100 tanh MLP with a diagonal Gaussian head and a
separate value branch, RLlib's defaults for a Box action space) runs in torch on the GPU that steps the
environments, so a rollout of T steps of N environments is T x (policy forward + libagx step) with every
tensor resident in HBM; only the finished batch is handed to the learner.

This module is plumbing around the stepper (torch for device memory and the MLP); it contains no physics.
"""
import math

import torch
from torch import nn


class GaussianMLPPolicy(nn.Module):
    """fcnet_hiddens [100, 100], tanh, outputs (mean, log_std) per action dimension; value function on its own
    branch (RLlib: vf_share_layers False).  Actions are sampled unclipped -- the env clips to [-1, 1] (env.py:188)."""

    def __init__(self, obs_dim, act_dim, hidden=(100, 100)):
        super().__init__()
        def mlp(out):
            layers, d = [], obs_dim
            for h in hidden:
                layers += [nn.Linear(d, h), nn.Tanh()]
                d = h
            return nn.Sequential(*layers, nn.Linear(d, out))
        self.pi, self.vf = mlp(2 * act_dim), mlp(1)
        self.act_dim = act_dim
        for m in self.modules():                      # RLlib's normc initialisation, small final policy layer
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight)
                m.weight.data *= 1.0 / m.weight.data.norm(dim=1, keepdim=True)
                nn.init.zeros_(m.bias)
        self.pi[-1].weight.data *= 0.01

    def forward(self, obs):
        out = self.pi(obs)
        return out[..., :self.act_dim], out[..., self.act_dim:].clamp(-20.0, 2.0), self.vf(obs).squeeze(-1)

    @torch.no_grad()
    def act(self, obs, generator=None):
        mean, log_std, value = self(obs)
        eps = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        action = mean + log_std.exp() * eps
        logp = (-0.5 * eps * eps - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        return action, logp, value

    def log_prob(self, obs, action):
        mean, log_std, value = self(obs)
        z = (action - mean) / log_std.exp()
        return (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1), value


def gae(rewards, values, dones, gamma=0.99, lam=0.95):
    """Generalised advantage estimation over a [T, N] rollout.  values: [T + 1, N] (bootstrap value last);
    dones[t] marks that step t ended its episode (the env has already been reset, so nothing is carried over)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(rewards[0])
    for t in range(T - 1, -1, -1):
        live = 1.0 - dones[t].to(rewards.dtype)
        delta = rewards[t] + gamma * values[t + 1] * live - values[t]
        last = delta + gamma * lam * live * last
        adv[t] = last
    return adv, adv + values[:-1]


@torch.no_grad()
def collect(env, policy, horizon, generator=None):
    """horizon steps of every environment of a FeedingJacoVecEnv under `policy`; returns device tensors
    obs [T, N, O], actions [T, N, A], logp / rewards / values [T, N], dones [T, N] (uint8), last_value [N],
    and the env's per-step info [T, N, 8].  Call env.reset() once before the first collect; the env auto-resets."""
    n, dev = env.n_envs, env.device
    buf = dict(obs=torch.empty((horizon, n, env.obs_dim), device=dev), actions=torch.empty((horizon, n, env.act_dim), device=dev),
               logp=torch.empty((horizon, n), device=dev), rewards=torch.empty((horizon, n), device=dev),
               values=torch.empty((horizon + 1, n), device=dev), dones=torch.empty((horizon, n), dtype=torch.uint8, device=dev),
               info=torch.empty((horizon, n, 8), device=dev))
    obs = env.obs
    for t in range(horizon):
        buf['obs'][t].copy_(obs)
        action, logp, value = policy.act(obs, generator)
        action = action.contiguous()
        buf['actions'][t], buf['logp'][t], buf['values'][t] = action, logp, value
        obs, rew, done, info = env.step(action)
        buf['rewards'][t].copy_(rew); buf['dones'][t].copy_(done); buf['info'][t].copy_(info)
    buf['values'][horizon] = policy(obs)[2]
    return buf
