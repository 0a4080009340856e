"""Host-side cost of the RLlib adapters: env-steps/s of the same 4096 FeedingJaco environments stepped (a) through
vec_env.step with the actions already on the device (the bench path), (b) through AgxVectorEnv.vector_step (RLlib's VectorEnv contract:
python lists of per-env arrays in and out, one pinned device-to-host copy per step), (c) ScratchItchPR2Human through AgxMultiAgentBatchEnv
(per-agent dictionaries).  Prints one JSON line.   python tools/gpu_rllib_overhead.py [n_envs] [steps]"""
import json, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from assistive_gym_amd.rllib import AgxVectorEnv, AgxMultiAgentBatchEnv, AgxPipelinedBatchEnv
from assistive_gym_amd.vec_env import FeedingJacoVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
out = {'n_envs': n, 'steps': K}
env = FeedingJacoVecEnv(n, pool_size=64, seed=1001); env.reset()
tape = torch.rand((K + 10, n, 7), device='cuda') * 2 - 1
for k in range(10): env.step(tape[k])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(10, K + 10): env.step(tape[k])
torch.cuda.synchronize(); out['vec_env_device_actions'] = n * K / (time.perf_counter() - t0); env.close()

v = AgxVectorEnv('FeedingJaco-v1', n, pool_size=64); v.vector_reset()
acts = np.random.RandomState(0).uniform(-1, 1, (n, 7)).astype(np.float32)
al = list(acts)
for k in range(10): v.vector_step(al)
t0 = time.perf_counter()
for k in range(K): v.vector_step(al)
out['rllib_vector_env'] = n * K / (time.perf_counter() - t0); v.close()

# (d) the same ids behind the asynchronous BaseEnv contract, two half-batches in flight: poll() hands out one half's finished step while the other
# half's kernels run.  `adapter only`: prebuilt action dictionaries, results not looked at (as (b)); `with a sampler stand-in`: every
# observation, reward and done flag is read and an action dictionary is built per round, as ray 1.x's _env_runner does
p2 = AgxPipelinedBatchEnv('FeedingJaco-v1', n, pool_size=64)
half = [{i: {'agent0': acts[i]} for i in range(0, n // 2)}, {i: {'agent0': acts[i]} for i in range(n // 2, n)}]
for k in range(24): p2.poll(); p2.send_actions(half[k & 1])
t0 = time.perf_counter()
for k in range(2 * K): p2.poll(); p2.send_actions(half[k & 1])
torch.cuda.synchronize(); out['rllib_pipelined_base_env_adapter_only'] = n * K / (time.perf_counter() - t0)
t0 = time.perf_counter(); tot = 0.0
for k in range(2 * K):
    obs, rew, done, info, _ = p2.poll()
    to_send = {}
    for i, ao in obs.items():
        r = rew[i]['agent0']; tot += r if r is not None else 0.0
        if done[i]['__all__']:
            ao = p2.try_reset(i)
        to_send[i] = {'agent0': acts[i]}
    p2.send_actions(to_send)
torch.cuda.synchronize(); out['rllib_pipelined_base_env_with_sampler_stand_in'] = n * K / (time.perf_counter() - t0); p2.stop()

m = AgxMultiAgentBatchEnv('ScratchItchPR2Human-v1', n, pool_size=64); m.poll()
ad = {i: {'robot': acts[i], 'human': np.zeros(10, np.float32)} for i in range(n)}
for k in range(5): m.send_actions(ad); m.poll()
t0 = time.perf_counter()
for k in range(K // 3): m.send_actions(ad); m.poll()
out['rllib_multi_agent_batch_env_scratchitch_coop'] = n * (K // 3) / (time.perf_counter() - t0); m.stop()
print(json.dumps(out))
