import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


# lean CPU suite: parametrisations that repeat a check for one more robot run only with AGX_FULL_TESTS=1 (the GPU suite covers every robot)
full = pytest.mark.skipif(not os.environ.get('AGX_FULL_TESTS'), reason='lean CPU suite: set AGX_FULL_TESTS=1 (the GPU suite covers this case)')


_GPU_SELECTED = [False]


def no_gpu(reason='no GPU visible'):
    """A GPU test found no device.  In a run that SELECTED the GPU tests (`-m gpu`, what the driver runs on the MI355X box) or with
    AGX_REQUIRE_GPU=1 that is a failure -- a broken driver or a wrong HIP_VISIBLE_DEVICES must not turn the parity suite into a
    green run that checked nothing; in any other run (the CPU suite, a bare `pytest tests`) the test is skipped."""
    if _GPU_SELECTED[0] or os.environ.get('AGX_REQUIRE_GPU') == '1':
        pytest.fail(reason + ' in a run that selected the GPU tests (-m gpu / AGX_REQUIRE_GPU=1)')
    pytest.skip(reason)


def pytest_configure(config):
    m = config.getoption('-m') or ''
    _GPU_SELECTED[0] = 'gpu' in m and 'not gpu' not in m
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def blob():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('feeding_jaco')


@pytest.fixture(scope='session')
def oracle(blob):
    from oracle_lib import Oracle
    return Oracle(blob)
