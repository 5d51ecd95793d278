import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/pointnet2_oracle.c) -- the checker, never the product."""
    from oracle import pointnet2_oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(os.path.join(GOLDEN, "pointnet2_ops.npz"))


@pytest.fixture(scope="session")
def golden_sa():
    return np.load(os.path.join(GOLDEN, "sa_module.npz"))


@pytest.fixture(scope="session")
def dev():
    import torch
    if os.environ.get("CODA_DEBUG_CPU_DEVICE") == "1":  # local debugging of pure-torch layers only
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU; use -m 'not gpu' on CPU-only hosts")
    return torch.device("cuda:0")
