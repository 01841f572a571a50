import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/): the checker, never the thing under test in -m gpu tests."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import d3p_amd._lib as L
    L.load()
    L.require_device()
    return torch.device("cuda:0")
