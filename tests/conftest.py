import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the three libraries exist (they are git-ignored build products)."""
    need = [os.path.join(ROOT, "matfac_amd", "libmfx.so"), os.path.join(ROOT, "matfac_amd", "libmfhost.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def has_gpu():
    import ctypes as C
    from matfac_amd import _lib
    n = C.c_int(0)
    _lib.load().mfx_device_count(C.byref(n))
    return n.value > 0
