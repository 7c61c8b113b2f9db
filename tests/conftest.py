import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); builds oracle/libhs_oracle.so if needed."""
    from oracle import hs_oracle
    hs_oracle.build()
    return hs_oracle


@pytest.fixture(scope="session")
def hs():
    """The product package; on a GPU box the HIP library must be the thing that runs."""
    import opticalflowhs_amd
    return opticalflowhs_amd


@pytest.fixture(scope="session")
def gpu_ok():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no GPU is visible")
    return True
