import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (test infrastructure) and, when absent, libg2ohip are built."""
    from oracle import oracle as O
    O.build()
    lib = os.path.join(ROOT, "openslam_g2o_amd", "lib", "libg2ohip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    yield
