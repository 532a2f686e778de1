import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    """One rsm context on cuda:0 / HIP device 0. Fails loudly if the HIP library or the GPU is missing."""
    from reconstruction_amd import Context
    c = Context(0)
    yield c
    c.close()
