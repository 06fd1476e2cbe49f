import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """The GPU parity tests record how much of each tolerance they used (tests/_tol.py)."""
    try:
        import _tol
        _tol.dump(os.path.join(ROOT, "gpurun_out", "tolerances_observed.json"))
    except Exception:
        pass
