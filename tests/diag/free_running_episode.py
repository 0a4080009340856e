"""Diagnostic (GPU box): free-running 200-step episodes, HIP stepper vs oracle from identical states and action tapes
(no re-synchronisation): how far do trajectories drift, do episode returns / food events agree?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.vec_env import build_reset_pool
from oracle_lib import Oracle
np.set_printoptions(precision=4, suppress=True, linewidth=200)
blob = ModelBlob.load(); o = Oracle(blob)
n, T = 16, 200
states = build_reset_pool(blob, n, 7001)
st = Stepper(blob, n); st.set_state(states)
rng = np.random.RandomState(5)
ref = states.copy()
ret_g = np.zeros(n); ret_o = np.zeros(n); drift = []
t0 = time.time()
for k in range(T):
    a = rng.uniform(-1, 1, (n, blob.act_dim)).astype(np.float32)
    obs, rew, done, info = st.step_host(a)
    ret_g += rew
    for i in range(n):
        o_obs, o_rew, o_done, o_info = o.step(ref[i], a[i]); ret_o[i] += o_rew
    if k % 25 == 24 or k == T - 1:
        got = st.get_state()
        dq = np.abs(blob.view(got)['q'] - blob.view(ref)['q']).max(1)
        drift.append((k + 1, float(np.median(dq)), float(dq.max())))
print('drift (step, median |dq|, max |dq|):', drift)
got = st.get_state(); vg, vr = blob.view(got), blob.view(ref)
print('returns gpu   ', ret_g); print('returns oracle', ret_o)
print('food_alive gpu   ', vg['food_alive']); print('food_alive oracle', vr['food_alive'])
print('task_success gpu', vg['task_success'], 'oracle', vr['task_success'])
print('corr', np.corrcoef(ret_g, ret_o)[0, 1], 'mean |dret|', np.abs(ret_g - ret_o).mean(), 'oracle time %.1f s' % (time.time() - t0))
