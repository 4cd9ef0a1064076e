import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pytorch-asr_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# the tests run small one-off convolution shapes: let MIOpen pick a solver from its
# heuristics instead of benchmarking every candidate on a fresh box (minutes)
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope='session')
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle
