import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "exemplar-vae_amd")          # drop-in root: models/, utils/, evae/, csrc/
for p in (PKG, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__)), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(params=["fp32-mfma", "x6"])
def gemm_pipe(request):
    """Run a test on both matrix pipes: the fp32-MFMA GEMM kernel, and the split-bf16 kernel (csrc/evae_gemm_x6.h) forced
    for every eligible launch whatever its row count (its default threshold would leave small test shapes on fp32)."""
    from evae import ops
    ops.gemm_x6_configure(1 if request.param == "x6" else 0, 0)
    yield request.param
    ops.gemm_x6_configure(1, 2048)
