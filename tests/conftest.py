import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle as orc
    orc.load()
    return orc


@pytest.fixture(scope="session")
def gpu():
    """torch device for GPU tests; fails (does not skip) if the HIP library is missing."""
    import torch
    from vulkanrenderer_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    return torch.device("cuda:0")
