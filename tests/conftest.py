import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-matching-webui_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(params=[1, 0], ids=["split-f16x3", "exact-f32"])
def precision(request):
    """Run a GPU test in both arithmetic modes of the matrix-core kernels."""
    import torch

    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, request.param)
    yield request.param
    backend.set_precision(dev, 1)


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (hipcc cross-compiles without a GPU)."""
    from imcui_hip import build, load_library

    build.build()
    return load_library()
