"""Loads libmfx.so (the HIP/C-ABI product).  Fails loudly when it is missing: there is
no Python or CPU fallback for any entry point."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFX_LIBRARY") or os.path.join(_HERE, "libmfx.so")   # MFX_LIBRARY: a diagnostic build (scripts/exp_bound.sh)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mfx.h")

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "matfac_amd: %s not found. Build it with `make -C matfac_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    return _lib
