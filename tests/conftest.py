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
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _library_options_back_to_defaults():
    """Schedule options are process-wide (cm_set_option): whatever a test forced -- directly or through the package's CM_* variables --
    is undone after it, so the next test starts from the library's own choices."""
    yield
    try:
        from cleanmarl_amd import _native as N
    except Exception:
        return
    if N._lib is not None:
        for key, default in N._DEFAULTS.items():
            N._lib.cm_set_option(key.encode(), default.encode())
        N._applied.clear()
        N._env_seen.clear()
