import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box: pytest -m gpu)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the REFERENCE's own torch functions on CPU (tests/golden/make_golden.py)."""
    return dict(np.load(os.path.join(GOLDEN, "reference_cpu.npz")))


@pytest.fixture(scope="session")
def golden_r2():
    """More outputs of the reference's own Python on CPU (tests/golden/make_golden_r2.py)."""
    return dict(np.load(os.path.join(GOLDEN, "reference_cpu_r2.npz")))


@pytest.fixture(scope="session")
def regression():
    """Oracle-generated vectors for the CUDA-only operators (parity unpinned)."""
    return dict(np.load(os.path.join(GOLDEN, "oracle_regression.npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu
    cpu.lib()
    return cpu


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


@pytest.fixture(scope="session")
def golden_r3():
    """Whole reference networks and the KDTree label transfer on CPU (tests/golden/make_golden_r3.py)."""
    return dict(np.load(os.path.join(GOLDEN, "reference_cpu_r3.npz")))
