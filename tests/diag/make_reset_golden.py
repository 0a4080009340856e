"""Writes tests/golden/reset_generator.npz: pre-settle state records of the reset generator for fixed seeds, produced by
the numpy restatement oracle/reset_oracle.py (NOT reference data -- the reference's reset cannot run here).  The fixture
freezes oracle and device together: a change that moves both in the same wrong direction still trips the test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import reset_oracle as ro
from assistive_gym_amd.blob import ModelBlob

blob = ModelBlob.load()
o = ro.with_collision_check(blob.words)
seeds = np.array([1001, 1002, 1003, 1004, (1 << 40) + 5, (1 << 63) + 12345], dtype=np.uint64)
modes = [(-1, -1)] * len(seeds) + [(3, 1), (1, 0)]
all_seeds = [int(s) for s in seeds] + [31, 32]
states, infos = [], []
for s, (imp, gen) in zip(all_seeds, modes):
    st, info = o.sample(s, imp, gen)
    states.append(st); infos.append([info['ik_ok'], info['ik_restarts'], info['ik_pos_err'], info['impairment']])
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'reset_generator.npz'), blob_version=blob.h['VERSION'],
                    seeds=np.array(all_seeds, dtype=np.uint64), impairment_mode=np.array([m[0] for m in modes]),
                    gender_mode=np.array([m[1] for m in modes]), states=np.array(states, dtype=np.float32), info=np.array(infos, dtype=np.float64))
print('wrote', len(states), 'records')
