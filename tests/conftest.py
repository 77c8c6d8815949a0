import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def torch_gpu_runtime_first():
    """Some GPU tests use PyTorch tensors beside the library (bench.py, the allocation-free entry point).  In one
    process PyTorch's HIP runtime has to come up before libcosmopol_hip.so brings up the system's: the other way
    round torch reports that no GPU is available."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield
