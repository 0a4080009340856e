"""libagx.so loads without a GPU, exports every symbol include/agx.h declares, and refuses to run."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from assistive_gym_amd.build import build
    build()
    from assistive_gym_amd import libagx
    return libagx.load()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, 'include', 'agx.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(agx_[a-z_]+)\s*\(', hdr))
    assert len(names) >= 20
    from assistive_gym_amd import libagx
    assert names == set(libagx.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_no_cpu_path(lib, blob):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = C.c_void_p()
    words = np.ascontiguousarray(blob.words)
    rc = lib.agx_create(words.ctypes.data_as(C.c_void_p), C.c_size_t(words.nbytes), 4, 0, C.byref(h))
    assert rc == -4 and b'no HIP device' in lib.agx_last_error()
    bad = words.copy(); bad[0] = 0
    # argument / blob validation happens before any device call
    assert lib.agx_create(bad.ctypes.data_as(C.c_void_p), C.c_size_t(bad.nbytes), 4, 0, C.byref(h)) == -2
    assert lib.agx_create(None, C.c_size_t(0), 4, 0, C.byref(h)) == -1
    assert lib.agx_lds_bytes_per_env() <= 64 * 1024
