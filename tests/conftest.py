import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    """Initialise device 0 or fail loudly (GPU tests never fall back to the CPU)."""
    from pyqsm_amd import _lib
    _lib.require_gpu(0)
    return 0
