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
from assistive_gym_amd.rllib import AgxVectorEnv, AgxMultiAgentBatchEnv
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

m = AgxMultiAgentBatchEnv('ScratchItchPR2Human-v1', n, pool_size=64); m.poll()
ad = {i: {'robot': acts[i], 'human': np.zeros(10, np.float32)} for i in range(n)}
for k in range(5): m.send_actions(ad); m.poll()
t0 = time.perf_counter()
for k in range(K // 3): m.send_actions(ad); m.poll()
out['rllib_multi_agent_batch_env_scratchitch_coop'] = n * (K // 3) / (time.perf_counter() - t0); m.stop()
print(json.dumps(out))
