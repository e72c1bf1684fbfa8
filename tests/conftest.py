import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both shared libraries must exist; build them if a fresh checkout has not yet."""
    import __graft_entry__ as g
    from oracle_binding import ORACLE_LIB
    from strolle_amd.api import LIB_PATH
    if not (os.path.exists(LIB_PATH) and os.path.exists(ORACLE_LIB)):
        g.build()
