"""How many concurrent row streams does the Gauss-Seidel sweep of a FeedingJaco substep offer?  

Runs the product kernel sources on the CPU wave emulator (variant 'feeding_trace_sched': csrc/agx_pgs_lvs.h records the DoF mask of
every row and, per sweep and part, the rows the no-op rule lets through) and list-schedules every part onto `width` lane groups:
a row goes to the earliest step after every EARLIER row of the part that shares a velocity slot with it (so any two rows that do not
commute keep their order: the schedule computes bit-for-bit what the sequential sweep computes).  Reported per policy:
  static   one schedule per substep over all rows; a step is executed when at least one of its rows is active
  dynamic  the active rows are re-scheduled whenever the mask of a part changes
Output: JSON with visits of today's kernel, steps of either policy, rows per step."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.host.reset import make_states
from emu_lib import Emu


def schedule(rows, masks, width):
    """rows: row indices in visit order; masks[r]: python int DoF mask -> step of every row, number of steps"""
    last = {}; fill = {}; step = {}; n = 0
    for r in rows:
        m = masks[r]; e = 0; b = 0; mm = m
        while mm:
            if mm & 1 and b in last: e = max(e, last[b] + 1)
            mm >>= 1; b += 1
        s = e
        while fill.get(s, 0) >= width: s += 1
        fill[s] = fill.get(s, 0) + 1; step[r] = s; n = max(n, s + 1)
        mm = m; b = 0
        while mm:
            if mm & 1: last[b] = s
            mm >>= 1; b += 1
    return step, n


def study(trace, width):
    out = dict(substeps=0, sweeps=0, rows=0, visits=0, static_steps=0, dynamic_steps=0, reschedules=0, static_full_steps=0)
    i = 0; n = len(trace)
    while i < n:
        assert trace[i] == -1
        R, nnc, nc = (int(x) for x in trace[i + 1:i + 4]); i += 4
        masks = []
        for r in range(R):
            lo, hi, m2 = (int(x) & 0xffffffff for x in trace[i:i + 3]); i += 3
            masks.append(lo | hi << 32 | m2 << 64)
        nA = nnc + nc
        stA, nsA = schedule(range(nA), masks, width); stF, nsF = schedule(range(nA, nA + nc), masks, width)
        out['substeps'] += 1; out['static_full_steps'] += nsA + nsF
        prev = {}; sup = {}; two = {}
        while i < n and trace[i] != -1:
            # one sweep: -2 (A rows 0..63), -3 (A rows 64..127), -4 (friction todo, lane = contact)
            a0 = (int(trace[i + 1]) & 0xffffffff) | (int(trace[i + 2]) & 0xffffffff) << 32
            a1 = (int(trace[i + 4]) & 0xffffffff) | (int(trace[i + 5]) & 0xffffffff) << 32
            td = (int(trace[i + 7]) & 0xffffffff) | (int(trace[i + 8]) & 0xffffffff) << 32
            assert trace[i] == -2 and trace[i + 3] == -3 and trace[i + 6] == -4
            i += 9
            actA = [r for r in range(nA) if ((a0 >> r) & 1 if r < 64 else (a1 >> (r - 64)) & 1)]
            actF = [nA + k for k in range(nc) if (td >> k) & 1]
            out['sweeps'] += 1; out['rows'] += R; out['visits'] += len(actA) + len(actF)
            out['static_steps'] += len({stA[r] for r in actA}) + len({stF[r] for r in actF})
            # policy 'superset': a schedule over a SET of rows serves every sweep whose active rows are a subset of it (steps without an
            # active row are dropped); it is rebuilt -- over the active rows -- only when a row outside the set turns up or when the set
            # has more than `slack` x the active rows
            # policy 'two': the static schedule serves the sweeps that visit every row of the part (the re-test sweeps); a second one, over the
            # rows active at the first other sweep, serves the rest and is rebuilt when a row outside its set turns up
            for key, act, full in (('A', actA, nA), ('F', actF, nc)):
                if len(act) == full: out['two_steps'] = out.get('two_steps', 0) + (nsA if key == 'A' else nsF); continue
                cur = two.get(key)
                if cur is None or not set(act) <= cur[0]:
                    stp, _ = schedule(act, masks, width); two[key] = (set(act), stp); out['two_reschedules'] = out.get('two_reschedules', 0) + 1
                out['two_steps'] = out.get('two_steps', 0) + len({two[key][1][r] for r in act})
            for key, act in (('A', actA), ('F', actF)):
                cur = sup.get(key)
                if cur is None or not set(act) <= cur[0] or len(cur[0]) > 1.25 * len(act) + 2:
                    stp, _ = schedule(act, masks, width); sup[key] = (set(act), stp); out['superset_reschedules'] = out.get('superset_reschedules', 0) + 1
                out['superset_steps'] = out.get('superset_steps', 0) + len({sup[key][1][r] for r in act})
            for key, act in (('A', actA), ('F', actF)):
                t = tuple(act)
                if prev.get(key, (None,))[0] != t:
                    prev[key] = (t, schedule(act, masks, width)[1]); out['reschedules'] += 1
                out['dynamic_steps'] += prev[key][1]
    return out


if __name__ == '__main__':
    blob = ModelBlob.load('feeding_jaco')
    emu = Emu(blob, 'feeding_trace_sched')
    tr = (C.c_int * (1 << 24)).in_dll(emu.L, 'g_sched_trace'); cnt = C.c_int.in_dll(emu.L, 'g_sched_n')
    nenv, nstep = int(os.environ.get('ENVS', 4)), int(os.environ.get('STEPS', 6))
    st, _ = make_states(blob, nenv, seed=1001)
    rng = np.random.RandomState(0)
    for e in range(nenv):
        s = st[e].copy(); emu.settle(s, 25)
        cnt.value = 0 if e == 0 else cnt.value
        if e == 0: cnt.value = 0
        for k in range(nstep):
            emu.step(s, rng.uniform(-1, 1, blob.act_dim).astype(np.float32))
    trace = np.ctypeslib.as_array(tr)[:cnt.value].copy()
    res = {}
    for width in (2, 4):
        o = study(trace, width)
        o['visits_per_sweep'] = o['visits'] / o['sweeps']; o['rows_per_sweep'] = o['rows'] / o['sweeps']
        o['static_steps_per_sweep'] = o['static_steps'] / o['sweeps']; o['dynamic_steps_per_sweep'] = o['dynamic_steps'] / o['sweeps']
        o['streams_static'] = o['visits'] / o['static_steps']; o['streams_dynamic'] = o['visits'] / o['dynamic_steps']
        o['reschedules_per_substep'] = o['reschedules'] / o['substeps']; o['superset_steps_per_sweep'] = o['superset_steps'] / o['sweeps']; o['superset_reschedules_per_substep'] = o['superset_reschedules'] / o['substeps']; o['two_steps_per_sweep'] = o['two_steps'] / o['sweeps']; o['two_reschedules_per_substep'] = o['two_reschedules'] / o['substeps']; o['streams_superset'] = o['visits'] / o['superset_steps']
        res['width_%d' % width] = o
    print(json.dumps(res, indent=1))
