import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # PyTorch bundles its own HIP runtime; the product links the system one.  Both live in one process in the tests that use
    # torch streams / tensors next to the C ABI: let torch's runtime open the device first (as bench.py does by importing torch
    # first) -- initialising it after the product has been seen to fail with hipErrorNoDevice.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from hsutil import Oracle
    return Oracle()
