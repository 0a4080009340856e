import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


# lean CPU suite: parametrisations that repeat a check for one more robot run only with AGX_FULL_TESTS=1 (the GPU suite covers every robot)
full = pytest.mark.skipif(not os.environ.get('AGX_FULL_TESTS'), reason='lean CPU suite: set AGX_FULL_TESTS=1 (the GPU suite covers this case)')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def blob():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('feeding_jaco')


@pytest.fixture(scope='session')
def oracle(blob):
    from oracle_lib import Oracle
    return Oracle(blob)
