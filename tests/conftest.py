import os
import sys

import pytest

# glibc writes the reason of an abort of its own (a failed heap check) to the terminal, which a test
# run on a GPU box does not have: send it to standard error, where the log keeps it.
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")

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


def pytest_sessionstart(session):
    # An abort (SIGABRT) of the test process names its origin: the checker library installs a handler
    # that writes the native stack of the aborting thread to standard error, then hands over to
    # Python's faulthandler (which pytest enabled before this hook).
    try:
        from oracle import oracle as orc
        orc.lib().oracle_install_abort_trace()
    except Exception as e:  # the tests that need the checker will say so themselves
        print(f"abort trace not installed: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
