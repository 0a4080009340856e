"""Generate tests/golden/<model>_oracle_traj.npz for one model of every task besides FeedingJaco (which has its own fixture,
make_golden.py) with the CPU oracle: a host-sampled post-reset state (no device: rigid 'drop' stand-in for the rag doll, the arm's
fall on the oracle), 12 random actions, the oracle's observations / rewards / final state.  Regression fixtures that freeze oracle and
device together -- NOT reference / PyBullet data (the reference cannot run in this environment).

    python tests/diag/make_golden_tasks.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from assistive_gym_amd.blob import ModelBlob   # noqa: E402
from oracle_lib import Oracle                  # noqa: E402

MODELS = [('bed_bathing_sawyer', False), ('scratch_itch_pr2', True), ('scratch_itch_jaco', False), ('arm_manipulation_sawyer', False),
          ('arm_manipulation_pr2', False), ('feeding_sawyer', False), ('bed_bathing_pr2', True)]


def state_of(b, seed):
    name = b.meta.get('name', '')
    from assistive_gym_amd.model import compiler as L
    if b.task_kind == L.TASK_FEEDING:
        from assistive_gym_amd.host.reset import make_states
        st = make_states(b, 1, seed=seed)[0]
        Oracle(b).settle(st[0], 25)
        return st[0]
    if b.task_kind == L.TASK_BED_BATHING:
        from assistive_gym_amd.host.reset_bed import make_states
        return make_states(b, 1, seed=seed)[0][0]
    if b.task_kind == L.TASK_SCRATCH_ITCH:
        from assistive_gym_amd.host.reset_scratch import make_states
        return make_states(b, 1, seed=seed)[0][0]
    from assistive_gym_amd.host.reset_arm import make_states
    fo = Oracle(b.set_param('HUMAN_GRAVITY_Z', -1.0))

    def fall(st, n):
        st = st.copy()
        for i in range(len(st)):
            fo.settle(st[i], n)
        return st
    return make_states(b, 1, seed=seed, arm_settler=fall)[0][0]


for model, coop in MODELS:
    b = ModelBlob.load(model)
    b = b.coop() if coop else b
    o = Oracle(b)
    s = state_of(b, 1001).copy()
    state0 = s.copy()
    actions = np.random.RandomState(1001).uniform(-1, 1, (12, b.act_dim)).astype(np.float32)
    obs, rew = [], []
    for a in actions:
        ob, r, d, info = o.step(s, a)
        obs.append(ob); rew.append(r)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', '%s%s_oracle_traj.npz' % (model, '_coop' if coop else '')), state0=state0, actions=actions,
                        obs=np.array(obs), reward=np.array(rew, dtype=np.float64), state_end=s, coop=coop)
    print(model, 'coop' if coop else '', 'return', round(float(sum(rew)), 4))
