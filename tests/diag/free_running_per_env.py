"""Diagnostic (GPU box): per-env joint deviation from the oracle after 5..25 free-running steps (the protocol of
tests/test_gpu_parity.py::test_free_running_episode_tracks_oracle), to tell a chaotic environment from a defect."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.vec_env import build_reset_pool
from oracle_lib import Oracle
blob = ModelBlob.load(); o = Oracle(blob)
n = 16
states = build_reset_pool(blob, n, 7001)
st = Stepper(blob, n); st.set_state(states)
rng = np.random.RandomState(5); ref = states.copy()
for k in range(25):
    a = rng.uniform(-1, 1, (n, blob.act_dim)).astype(np.float32)
    st.step_host(a)
    for i in range(n):
        o.step(ref[i], a[i])
    if k % 5 == 4:
        dq = np.abs(blob.view(st.get_state())['q'] - blob.view(ref)['q']).max(1)
        print('step', k + 1, ' '.join('%.1e' % x for x in dq))
