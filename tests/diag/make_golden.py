"""Generate tests/golden/feeding_jaco_oracle_traj.npz with the CPU oracle (regression fixture;
NOT reference/PyBullet data -- the reference cannot run in this environment)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.host.reset import make_states
from oracle_lib import Oracle

blob = ModelBlob.load()
o = Oracle(blob)
st, _ = make_states(blob, 1, seed=1001)
s = st[0].copy()
o.settle(s, 25)
state0 = s.copy()
rng = np.random.RandomState(1001)
actions = rng.uniform(-1, 1, (20, blob.act_dim)).astype(np.float32)
obs, rew = [], []
for a in actions:
    ob, r, d, info = o.step(s, a)
    obs.append(ob); rew.append(r)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'feeding_jaco_oracle_traj.npz'), state0=state0, actions=actions,
                    obs=np.array(obs), reward=np.array(rew, dtype=np.float64), state_end=s)
print('wrote golden trajectory, return', sum(rew))
