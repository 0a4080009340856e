"""GPU box: cost of the device-side reset generator (agx_sample_reset + 25 settle substeps) at 4096 envs and
the throughput of FeedingJacoVecEnv with fresh per-episode resets vs the fixed pool."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.vec_env import FeedingJacoVecEnv

blob = ModelBlob.load()
n = 4096
out = {}
st = Stepper(blob, n)
info = torch.zeros((n, 4), device='cuda')
st.sample_reset(1); st.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
s = torch.cuda.current_stream().cuda_stream
ev[0].record(); st.sample_reset(1001, ik_info=info, stream=s); ev[1].record(); st.settle(25, s); ev[2].record()
torch.cuda.synchronize()
out['sample_ms'] = ev[0].elapsed_time(ev[1]); out['settle25_ms'] = ev[1].elapsed_time(ev[2])
out['ik_ok_frac'] = float(info[:, 0].mean()); out['ik_restarts_mean'] = float(info[:, 1].mean()); out['ik_restarts_max'] = float(info[:, 1].max())
st.close()
for mode in ('pool', 'device', 'pool', 'device'):
    env = FeedingJacoVecEnv(n, seed=1001, reset=mode)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    K = 600
    tape = torch.rand((50, n, 7), device='cuda', generator=g) * 2 - 1
    for k in range(50): env.step(tape[k % 50])
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(K): env.step(tape[k % 50])
    torch.cuda.synchronize(); dt = time.time() - t0
    out.setdefault('env_steps_per_s_' + mode, []).append(n * K / dt)
    out['mean_reward_' + mode] = float(env.reward.mean())
    env.close()
from assistive_gym_amd.rollout import GaussianMLPPolicy, collect, gae
env = FeedingJacoVecEnv(n, seed=1001, reset='device')
env.reset()
pi = GaussianMLPPolicy(env.obs_dim, env.act_dim).to(env.device)
collect(env, pi, 20)
torch.cuda.synchronize(); t0 = time.time()
T = 400
buf = collect(env, pi, T)
adv, ret = gae(buf['rewards'], buf['values'], buf['dones'])
torch.cuda.synchronize(); dt = time.time() - t0
out['rollout_env_steps_per_s_policy_in_loop_fresh_resets'] = n * T / dt
env.close()
print(json.dumps(out))
