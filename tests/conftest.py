import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only; skipped elsewhere)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native pieces exist (compiles them if this checkout has none yet)."""
    import rt_host
    import oracle_util
    if not os.path.exists(rt_host.LIB_PATH) or not os.path.exists(rt_host.TEST_LIB_PATH) or not os.path.exists(oracle_util.C_ORACLE_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return True
