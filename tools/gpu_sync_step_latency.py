"""GPU box: what does ONE env step cost when nothing else is in flight?  (The RLlib VectorEnv contract is synchronous: every vector_step starts on an idle GPU.)
For 4096 FeedingJaco environments: (a) the device-resident loop (steps enqueued back to back, one synchronise at the end), (b) the same loop with a
synchronise after every step, (c) with a synchronise and a host pause of 0.5 ms after every step (what a sampler does between steps), (d) agx_step_timed:
HIP events after every launch -- the summed kernel durations of an isolated step.  Prints one JSON line.  usage: python tools/gpu_sync_step_latency.py [steps]"""
import json, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from assistive_gym_amd.vec_env import FeedingJacoVecEnv
K = int(sys.argv[1]) if len(sys.argv) > 1 else 150
out = {}
for chunks in ('3', '1'):
    os.environ['AGX_CHUNKS'] = chunks
    env = FeedingJacoVecEnv(4096, pool_size=64, seed=1001); env.reset()
    tape = torch.rand((K, 4096, 7), device='cuda') * 2 - 1
    for k in range(20): env.step(tape[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K): env.step(tape[k])
    torch.cuda.synchronize(); back_to_back = (time.perf_counter() - t0) / K * 1e3
    t0 = time.perf_counter()
    for k in range(K): env.step(tape[k]); torch.cuda.synchronize()
    synced = (time.perf_counter() - t0) / K * 1e3
    t0 = time.perf_counter(); paused = 0.0
    for k in range(K):
        env.step(tape[k]); torch.cuda.synchronize()
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.5e-3: pass
        paused += time.perf_counter() - t1
    with_pause = ((time.perf_counter() - t0) - paused) / K * 1e3
    ms = [0.0, 0.0, 0.0]
    s = torch.cuda.current_stream().cuda_stream
    for k in range(20):
        m, c = env.stepper.step_timed(tape[k], env.obs, env.reward, env.done, env.info, s)
        ms = [a + b for a, b in zip(ms, m)]
    out['chunks_' + chunks] = dict(ms_per_step_back_to_back=round(back_to_back, 3), ms_per_step_synchronised=round(synced, 3), ms_per_step_synchronised_after_a_0p5ms_host_pause=round(with_pause, 3),
                                   isolated_step_kernel_ms_summed=dict(zip(('build', 'solve', 'finish'), [round(x / 20, 3) for x in ms])))
    env.close()
print(json.dumps(out))
