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

if len(sys.argv) > 1 and sys.argv[1] == '--half':        # child of (e): one synchronous VectorEnv of sys.argv[2] environments; prints its env-steps/s over a window both children share roughly
    nh, Kh, h = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    v = AgxVectorEnv('FeedingJaco-v1', nh, pool_size=64, env_offset=h * nh); v.vector_reset()
    th = np.random.RandomState(h).uniform(-1, 1, (Kh + 30, nh, 7)).astype(np.float32)
    tl = [list(t) for t in th]
    for j in range(30): v.vector_step(tl[j])
    t0 = time.perf_counter()
    for j in range(30, Kh + 30): v.vector_step(tl[j])
    print(nh * Kh / (time.perf_counter() - t0)); v.close(); sys.exit(0)
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
# a NEW random action per environment and step, as in (a) -- rounds 4-5 fed the SAME action vector at every step, which drives every arm into its joint limits,
# the table and the wheelchair within a few dozen steps: a heavier workload (11.6 ms per step on the GPU against 6.7), not a slower adapter
tape_host = np.random.RandomState(1).uniform(-1, 1, (K + 10, n, 7)).astype(np.float32)
tape_lists = [list(t) for t in tape_host]
for k in range(10): v.vector_step(tape_lists[k])
t0 = time.perf_counter()
for k in range(10, K + 10): v.vector_step(tape_lists[k])
out['rllib_vector_env'] = n * K / (time.perf_counter() - t0)
t0 = time.perf_counter()
for k in range(K): v.vector_step(al)
out['rllib_vector_env_same_action_every_step_as_measured_in_rounds_4_5'] = n * K / (time.perf_counter() - t0)
v.vector_reset()
hs = v.host_seconds; v.host_seconds = np.zeros(4)
for k in range(10, K + 10): v.vector_step(tape_lists[k])
out['rllib_vector_env_ms_per_step'] = dict(zip(('actions_to_device', 'enqueue_step_pack_copy', 'wait_for_the_gpu', 'results_to_python'), (v.host_seconds / K * 1e3).round(3).tolist()))
os.environ['AGX_CHUNKS'] = '1'
v1 = AgxVectorEnv('FeedingJaco-v1', n, pool_size=64); v1.vector_reset()
for k in range(10): v1.vector_step(tape_lists[k])
t0 = time.perf_counter()
for k in range(10, K + 10): v1.vector_step(tape_lists[k])
out['rllib_vector_env_one_chunk'] = n * K / (time.perf_counter() - t0); v1.close(); del os.environ['AGX_CHUNKS']
v.close()

# (e) two VectorEnvs of n / 2 environments (RLlib: num_workers = 2, rllib.worker_env), each stepped synchronously -- from two threads of this
# process (the GIL is released while a thread waits for its GPU results, so one worker's kernels run while the other is in its Python), and
# from two processes (what RLlib's rollout workers are)
import threading
def _worker(v, steps, lo, hi, out, k):
    tl = [list(t[lo:hi]) for t in tape_host]
    for j in range(10): v.vector_step(tl[j])
    bar.wait()
    t0 = time.perf_counter()
    for j in range(10, steps + 10): v.vector_step(tl[j])
    out[k] = time.perf_counter() - t0
vs = [AgxVectorEnv('FeedingJaco-v1', n // 2, pool_size=64, env_offset=h * (n // 2)) for h in range(2)]
for v in vs: v.vector_reset()
bar = threading.Barrier(2); el = [0.0, 0.0]
th = [threading.Thread(target=_worker, args=(vs[h], K, h * (n // 2), (h + 1) * (n // 2), el, h)) for h in range(2)]
for t in th: t.start()
for t in th: t.join()
out['rllib_two_vector_envs_two_threads'] = n * K / max(el)
for v in vs: v.close()
import subprocess
cmd = [sys.executable, os.path.abspath(__file__), '--half', str(n // 2), str(K)]
t0 = time.perf_counter()
ps = [subprocess.Popen(cmd + [str(h)], stdout=subprocess.PIPE, text=True) for h in range(2)]
rates = [float(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
out['rllib_two_vector_envs_two_processes'] = sum(rates)

# (d) the same ids behind the asynchronous BaseEnv contract, two half-batches in flight: poll() hands out one half's finished step while the other
# half's kernels run.  `adapter only`: prebuilt action dictionaries, results not looked at (as (b)); `with a sampler stand-in`: every
# observation, reward and done flag is read and an action dictionary is built per round, as ray 1.x's _env_runner does
p2 = AgxPipelinedBatchEnv('FeedingJaco-v1', n, pool_size=64)
def _half(k):
    h = k & 1; t = tape_host[(k >> 1) % (K + 10)]
    return {i: {'agent0': t[i]} for i in range(h * (n // 2), (h + 1) * (n // 2))}
pre = [_half(k) for k in range(2 * K)]
for k in range(24): p2.poll(); p2.send_actions(pre[k])
t0 = time.perf_counter()
for k in range(2 * K): p2.poll(); p2.send_actions(pre[k])
torch.cuda.synchronize(); out['rllib_pipelined_base_env_adapter_only'] = n * K / (time.perf_counter() - t0)
t0 = time.perf_counter(); tot = 0.0
for k in range(2 * K):
    obs, rew, done, info, _ = p2.poll()
    to_send = {}
    for i, ao in obs.items():
        r = rew[i]['agent0']; tot += r if r is not None else 0.0
        if done[i]['__all__']:
            ao = p2.try_reset(i)
        to_send[i] = {'agent0': tape_host[k % (K + 10)][i]}
    p2.send_actions(to_send)
torch.cuda.synchronize(); out['rllib_pipelined_base_env_with_sampler_stand_in'] = n * K / (time.perf_counter() - t0); p2.stop()

m = AgxMultiAgentBatchEnv('ScratchItchPR2Human-v1', n, pool_size=64); m.poll()
hr = np.random.RandomState(2).uniform(-1, 1, (8, n, 10)).astype(np.float32)
ads = [{i: {'robot': tape_host[j][i], 'human': hr[j][i]} for i in range(n)} for j in range(8)]      # (new random actions every step, cycling through eight prebuilt dictionaries)
for k in range(5): m.send_actions(ads[k % 8]); m.poll()
t0 = time.perf_counter()
for k in range(K // 3): m.send_actions(ads[k % 8]); m.poll()
out['rllib_multi_agent_batch_env_scratchitch_coop'] = n * (K // 3) / (time.perf_counter() - t0); m.stop()
print(json.dumps(out))
