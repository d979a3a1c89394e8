import os
import sys

import pytest

# PyTorch ships its own copy of the HIP runtime.  A process that uses both torch and
# libspgemm_amd.so must load torch first: the library then binds to the runtime torch brought; the
# other way round two runtimes end up in the process and torch finds no GPU.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
