"""The list scheduler of the wide row-local sweep (csrc/agx_pgs_lvw.h lvw_schedule, run on the CPU wave emulator): for any sequence of rows with slot masks the
schedule must (1) hold every row exactly once, (2) put at most four rows into a step, none of which share a slot, and (3) keep the ORIGINAL ORDER of every two
rows that share a slot (the earlier row in an earlier step) -- then visiting the steps in order, the rows of a step at once, computes bit for bit what the
sequential Gauss-Seidel sweep computes (rows without a common slot commute).  Also: the step count against the dependency chain, and the refusal beyond 64 steps."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope='module')
def sched():
    from emu_lib import lib
    L = lib(0)
    L.agx_emu_lvw_schedule.restype = C.c_int
    dummy = L.agx_emu_lvw_dummy_row()

    def run(masks):
        m3 = np.zeros((len(masks), 3), dtype=np.uint32)
        for r, m in enumerate(masks):
            m3[r] = [m & 0xffffffff, (m >> 32) & 0xffffffff, (m >> 64) & 0xffffffff]
        ss = np.zeros(64, dtype=np.uint32)
        n = L.agx_emu_lvw_schedule(m3.ctypes.data_as(C.c_void_p), C.c_int(len(masks)), ss.ctypes.data_as(C.c_void_p))
        steps = [[int((w >> (8 * j)) & 255) for j in range(4) if int((w >> (8 * j)) & 255) != dummy] for w in ss]
        return n, steps
    return run


def _check(masks, n, steps):
    assert n >= 0
    assert all(not s for s in steps[n:]) and all(steps[k] for k in range(n) if k == n - 1)
    where = {}
    for k, rows in enumerate(steps[:n]):
        assert len(rows) <= 4
        for r in rows:
            assert r not in where and 0 <= r < len(masks)
            where[r] = k
        for i in range(len(rows)):
            for j in range(i + 1, len(rows)):
                assert masks[rows[i]] & masks[rows[j]] == 0, ('rows of one step share a slot', k, rows)
    assert sorted(where) == list(range(len(masks)))
    for a in range(len(masks)):
        for b in range(a + 1, len(masks)):
            if masks[a] & masks[b]:
                assert where[a] < where[b], ('order of two rows that share a slot', a, b, where[a], where[b])
    # the critical path: longest chain of rows each sharing a slot with an earlier one
    depth = []
    for b in range(len(masks)):
        depth.append(1 + max([depth[a] for a in range(b) if masks[a] & masks[b]] + [0]))
    return max(depth + [0])


def test_feeding_like_scene(sched):
    """ten motor rows on the robot block, four on the person's, six rows of the tool constraint (robot + spoon), then contacts: food x spoon, food x food, bowl x table"""
    rng = np.random.RandomState(0)
    robot, human = (1 << 10) - 1, ((1 << 4) - 1) << 10
    body = lambda b: ((1 << 6) - 1) << (14 + 6 * int(b))
    for trial in range(20):
        masks = [robot] * 10 + [human] * 4 + [robot | body(0)] * 6
        for c in range(rng.randint(30, 56)):
            kind = rng.rand()
            if kind < 0.5:
                masks.append(body(0) | body(2 + rng.randint(8)))
            elif kind < 0.75:
                a, b = rng.choice(8, 2, replace=False); masks.append(body(2 + a) | body(2 + b))
            else:
                masks.append(body(1))
        n, steps = sched(masks)
        chain = _check(masks, n, steps)
        assert chain <= n <= chain + 4 and n < len(masks)          # as short as the dependency chain allows (greedy: a few steps above it at most)


def test_random_masks_and_extremes(sched):
    rng = np.random.RandomState(1)
    for trial in range(30):
        nr = rng.randint(1, 128)
        masks = [int(rng.randint(1, 1 << 12)) << int(rng.randint(0, 80)) for _ in range(int(nr))]
        n, steps = sched(masks)
        if n >= 0:
            _check(masks, n, steps)
        else:
            depth = []
            for b in range(nr):
                depth.append(1 + max([depth[a] for a in range(b) if masks[a] & masks[b]] + [0]))
            assert max(depth) > 60 or nr > 4 * 60          # refused only when the chain (or the sheer number of rows) does not fit 64 steps
    # independent rows: four per step
    n, steps = sched([1 << k for k in range(90)])
    assert n == 23 and _check([1 << k for k in range(90)], n, steps) == 1
    # a single chain longer than 64 steps is refused (the caller takes the narrow sweep)
    assert sched([1] * 65)[0] == -1
    n, steps = sched([1] * 64)
    assert n == 64 and _check([1] * 64, n, steps) == 64
    # rows without pairs conflict with nothing
    n, steps = sched([0, 0, 0, 0, 0])
    assert n == 2
