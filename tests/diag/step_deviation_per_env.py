"""Diagnostic (GPU box): per-env single-step deviations of the HIP stepper from the oracle (same protocol as
tests/test_gpu_parity.py::test_step_matches_oracle), printing the environments that exceed the tolerance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.vec_env import build_reset_pool
from oracle_lib import Oracle
np.set_printoptions(precision=6, suppress=True, linewidth=220)
blob = ModelBlob.load(); oracle = Oracle(blob)
n, steps = 32, 6
states = build_reset_pool(blob, n, 5001)
st = Stepper(blob, n)
rng = np.random.RandomState(7)
ref = states.copy()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
for k in range(steps):
    st.set_state(ref)
    before = ref.copy()
    actions = rng.uniform(-1, 1, (n, blob.act_dim)).astype(np.float32)
    obs, rew, done, info = st.step_host(actions)
    got = st.get_state()
    for i in range(n):
        o_obs, o_rew, o_done, o_info = oracle.step(ref[i], actions[i])
        vg, vr, vb = blob.view(got[i]), blob.view(ref[i]), blob.view(before[i])
        dq = np.abs(vg['q'] - vr['q'])[0]
        if dq.max() > 3e-4:
            print('step', k, 'env', i, 'dq', dq, 'rows gpu/oracle', info[i, 7], o_info[7], 'ncon', info[i, 6], o_info[6])
            print('   q before', vb['q'][0]); print('   q oracle', vr['q'][0]); print('   q gpu   ', vg['q'][0])
            print('   qt oracle', vr['qt'][0]); print('   qt gpu   ', vg['qt'][0])
            print('   action', actions[i])
            np.save(os.path.join(ROOT, 'gpurun_out', 'dev_state_%d_%d.npy' % (k, i)), before[i]); np.save(os.path.join(ROOT, 'gpurun_out', 'dev_action_%d_%d.npy' % (k, i)), actions[i])
print('done')
