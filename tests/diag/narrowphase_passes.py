"""Narrowphase statistics of settled FeedingJaco states on the CPU wave emulator (instrumented build, -DAGX_EMU_TRACE_GJK): per
gjk_distance call (= one 64-lane pass of collide_flush) the lanes' iteration counts and hull sizes.  A pass costs what its slowest
lane costs: sum over passes of max(iterations) against the mean.  Not a test; run by hand:  python tests/diag/narrowphase_passes.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.host.reset import make_states
from oracle_lib import Oracle
from emu_lib import Emu, lib


def main(n_states=6, steps=3):
    b = ModelBlob.load('feeding_jaco')
    L = lib('feeding_trace')
    e = Emu(b, kind='feeding_trace')
    states, _ = make_states(b, n_states, seed=4242)
    o = Oracle(b)
    for i in range(n_states):
        o.settle(states[i], 25)              # the pool's settle: the food rests on the spoon
    rng = np.random.RandomState(5)
    tr = (C.c_int * (1 << 22)).in_dll(L, 'g_gjk_trace'); n = C.c_int.in_dll(L, 'g_gjk_n')
    rows = []
    for i in range(n_states):
        s = states[i].copy()
        for k in range(steps):
            n.value = 0
            e.step(s, rng.uniform(-1, 1, b.act_dim).astype(np.float32))
            t = np.frombuffer(tr, dtype=np.int32, count=n.value).reshape(-1, 9).copy()
            rows.append(t)
            calls = np.unique(t[:, 0])
            per = []
            for c in calls:
                q = t[t[:, 0] == c]
                per.append((len(q), q[:, 2].max(), q[:, 2].mean(), (q[:, 3] + q[:, 4]).max(), (q[:, 3] + q[:, 4]).mean(), int((q[:, 3] == 1).sum())))
            per = np.array(per)
            print('state %d step %d: %d passes in 5 substeps, pairs %d, sum of max iterations %d, sum of mean iterations %.1f' %
                  (i, k, len(per), per[:, 0].sum(), per[:, 1].sum(), per[:, 2].sum()))
            if i == 0 and k == 0:
                for p in per[:12]:
                    print('   pass: lanes %3d  max it %2d  mean it %.2f  max verts %3d  mean verts %.1f  single-vertex A %3d' % tuple(p))
    t = np.concatenate(rows)
    print('all pairs: iterations histogram', np.bincount(t[:, 2]))
    pt = t[(t[:, 3] == 1) | (t[:, 4] == 1)]
    print('pairs with a one-vertex core (spheres): %d of %d, iterations histogram' % (len(pt), len(t)), np.bincount(pt[:, 2]))
    ot = t[(t[:, 3] != 1) & (t[:, 4] != 1)]
    print('other pairs: iterations histogram', np.bincount(ot[:, 2]), 'hull sizes |A|+|B| median', np.median(ot[:, 3] + ot[:, 4]))
    print('sphere pairs: stopped by the separating-axis early-out %d of %d; limit (um) median %d; core distance of the others (um): median %d, share beyond the limit %.2f' % ((pt[:, 6] == 1).sum(), len(pt), np.median(pt[:, 8]), np.median(pt[pt[:, 6] == 0][:, 7]), (pt[pt[:, 6] == 0][:, 7] > pt[pt[:, 6] == 0][:, 8]).mean()))
    print('sphere pairs: partner hull size median %d max %d' % (np.median(pt[:, 3] + pt[:, 4] - 1), (pt[:, 3] + pt[:, 4] - 1).max()))


if __name__ == '__main__':
    main()
