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
def golden_detector():
    return np.load(os.path.join(GOLDEN, "detector.npz"))


@pytest.fixture(scope="session", params=["default", "glibc"])
def golden_polar(request):
    return request.param, np.load(os.path.join(GOLDEN, f"polar_{request.param}.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from echoseal_amd.engine import RxEngine
    return RxEngine(0)
