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


def pytest_sessionfinish(session, exitstatus):
    """Observed maxima of the parity metrics (tests/parity.py) of this session -> gpurun_out/parity_observed.txt (or $CM_PARITY_REPORT):
    the evidence that the bars of the GPU tests are bars the kernels pass with room, and by how much."""
    try:
        import parity
    except Exception:
        return
    if not parity.OBSERVED:
        return
    path = os.environ.get("CM_PARITY_REPORT") or os.path.join(ROOT, "gpurun_out", "parity_observed.txt")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        parity.report(path)
    except OSError:
        pass
