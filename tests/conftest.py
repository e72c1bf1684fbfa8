import os
import sys

import pytest

try:   # one HIP runtime per process: torch's bundled one must be the first (strolle_amd/api.py load_library)
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both shared libraries are (re)built from source before any test runs: make is incremental and its dependency
    files cover every header, so a stale libstrolle_hip.so can never be what the suite passes against."""
    import __graft_entry__ as g
    g.build()
