"""SURVEY 5 (race / memory checking of the test infrastructure): the CPU oracle -- the judge of every parity test -- built with
AddressSanitizer + UndefinedBehaviorSanitizer and run on states of every task (incl. a garment); the kernel sources of the wave
emulator built with UndefinedBehaviorSanitizer (its fibres do not mix with ASan's stack bookkeeping) and stepped once."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer', '-g', '-O1']


@pytest.fixture(scope='module')
def oracle_san(tmp_path_factory):
    d = tmp_path_factory.mktemp('san')
    exe = str(d / 'oracle_sanitize')
    subprocess.check_call(['gcc', '-std=gnu11'] + SAN + ['-o', exe, os.path.join(ROOT, 'tests', 'diag', 'oracle_sanitize_main.c'), os.path.join(ROOT, 'oracle', 'agx_oracle.c'), '-lm'])
    return exe, d


def _states(model, n, coop=False):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.model import compiler as L
    b = ModelBlob.load(model)
    b = b.coop() if coop else b
    mod = {L.TASK_FEEDING: 'reset', L.TASK_BED_BATHING: 'reset_bed', L.TASK_SCRATCH_ITCH: 'reset_scratch', L.TASK_ARM_MANIPULATION: 'reset_arm', L.TASK_DRESSING: 'reset_dressing'}[b.task_kind]
    ms = __import__('assistive_gym_amd.host.' + mod, fromlist=['make_states']).make_states
    out = ms(b, n, seed=8801)
    return b, out[0], (out[1] if b.task_kind == L.TASK_DRESSING else None)


@pytest.mark.parametrize('model,coop,steps', [('feeding_jaco', True, 6), ('bed_bathing_sawyer', False, 4), ('scratch_itch_pr2', True, 4), ('arm_manipulation_pr2', False, 3), ('dressing_baxter', False, 1)])
def test_oracle_under_asan_and_ubsan(oracle_san, model, coop, steps):
    exe, d = oracle_san
    b, st, cloth = _states(model, 2, coop)
    bp, sp = str(d / (model + '.blob')), str(d / (model + '.states'))
    b.words.tofile(bp); np.ascontiguousarray(st, dtype=np.float32).tofile(sp)
    args = [exe, bp, sp, str(steps)]
    if cloth is not None:
        cp = str(d / (model + '.cloth')); np.ascontiguousarray(cloth, dtype=np.float32).tofile(cp); args.append(cp)
    r = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
    assert r.returncode == 0 and r.stdout.startswith('ok'), (r.stdout[-500:], r.stderr[-3000:])


def test_emulator_kernel_sources_under_ubsan(tmp_path):
    """the product kernel sources (csrc/*.h) on the wave emulator, compiled with -fsanitize=undefined: one env.step of FeedingJaco"""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    so = str(tmp_path / 'libagx_emu_ubsan.so')
    subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-fsanitize=undefined', '-fno-sanitize-recover=undefined', '-I' + os.path.join(ROOT, 'tests', 'emu'),
                           '-I' + os.path.join(ROOT, 'assistive_gym_amd', 'csrc'), '-o', so, os.path.join(ROOT, 'tests', 'emu', 'emu_main.cpp')])
    code = '''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.host.reset import make_states
b = ModelBlob.load('feeding_jaco')
L = C.CDLL(%r); L.agx_emu_run.restype = C.c_int
st, _ = make_states(b, 1, seed=8802)
p = lambda a: a.ctypes.data_as(C.c_void_p)
w = np.ascontiguousarray(b.words); s = st[0].copy()
obs = np.zeros(b.obs_dim, np.float32); rew = np.zeros(1, np.float32); done = np.zeros(4, np.uint8); info = np.zeros(8, np.float32); act = np.full(b.act_dim, 0.3, np.float32)
assert L.agx_emu_run(p(w), p(s), p(act), p(obs), p(rew), p(done), p(info), None, C.c_int(1), C.c_int(2)) == 0
assert L.agx_emu_run(p(w), p(s), p(act), p(obs), p(rew), p(done), p(info), None, C.c_int(0), C.c_int(0)) == 0
assert np.isfinite(obs).all()
print('ok')
''' % (ROOT, os.path.join(ROOT, 'tests'), so)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1'))
    assert r.returncode == 0 and 'ok' in r.stdout, (r.stdout[-300:], r.stderr[-3000:])
