"""Builds and binds the CPU wave-emulator build of the product kernel sources (tests/emu/)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu')
CSRC = os.path.join(ROOT, 'assistive_gym_amd', 'csrc')
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(EMU, 'libagx_emu.so')
        deps = [os.path.join(EMU, f) for f in ('emu_main.cpp', 'agx_wave.h')] + \
               [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')] + [os.path.join(ROOT, 'include', 'agx_blob.h')]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-I' + EMU, '-I' + CSRC, '-o', so,
                                   os.path.join(EMU, 'emu_main.cpp')])
        _LIB = C.CDLL(so)
        _LIB.agx_emu_run.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Emu:
    DBG_HDR = 16 + 64 * 16 + 256
    DBG_LAM = DBG_HDR + 160 * 10
    DBG_TIME = DBG_LAM + 160
    DEBUG_WORDS = DBG_TIME + 16

    def __init__(self, blob):
        self.blob = blob
        self.L = lib()
        self.words = np.ascontiguousarray(blob.words)

    def _run(self, state, action, mode, nsettle, debug=False):
        obs = np.zeros(self.blob.obs_dim, dtype=np.float32)
        rew = np.zeros(1, dtype=np.float32)
        done = np.zeros(4, dtype=np.uint8)
        info = np.zeros(8, dtype=np.float32)
        dbg = np.zeros(self.DEBUG_WORDS, dtype=np.float32) if debug else None
        act = np.ascontiguousarray(action if action is not None else np.zeros(self.blob.act_dim), dtype=np.float32)
        rc = self.L.agx_emu_run(_p(self.words), _p(state), _p(act), _p(obs), _p(rew), _p(done), _p(info), _p(dbg), C.c_int(mode), C.c_int(nsettle))
        assert rc == 0, 'wave emulator reported divergent control flow'
        return obs, float(rew[0]), bool(done[0]), info, dbg

    def step(self, state, action, debug=False):
        return self._run(state, action, 0, 0, debug)

    def settle(self, state, n):
        self._run(state, None, 1, n)

    def sample(self, seed, impairment_mode=-1, gender_mode=-1):
        """device-side reset generator (csrc/agx_reset.h) for one env -> (state record, info[4])"""
        st = np.zeros(self.blob.state_words, dtype=np.float32)
        info = np.zeros(4, dtype=np.float32)
        rc = self.L.agx_emu_sample(_p(self.words), _p(st), C.c_uint64(seed), C.c_int(impairment_mode), C.c_int(gender_mode), _p(info))
        assert rc == 0, 'wave emulator reported divergent control flow'
        return st, info

    def observe(self, state):
        return self._run(state, None, 2, 0)[0]
