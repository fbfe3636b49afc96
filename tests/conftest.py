import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_native():
    from oracle import native

    native.load()
    return native


@pytest.fixture(scope="session")
def gpu():
    """HIP device + loaded libcorsair_hip.so; fails loudly when either is missing."""
    import torch

    from corsair_amd import _lib

    _lib.load()
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.require_gpu()
    return torch.device("cuda:0")
