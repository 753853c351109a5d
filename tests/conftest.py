import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library wrapper; GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from optbayesexpt_amd import _lib
    return _lib.load()
