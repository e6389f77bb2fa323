import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def gpu():
    """Handle + torch device for the -m gpu tests; fails loudly without the native library."""
    import torch
    from spgpu_amd import capi  # raises if libspgpu.so is missing
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    handle = capi.create_handle(0)
    yield handle
    torch.cuda.synchronize()
    capi.spgpuDestroy(handle)
