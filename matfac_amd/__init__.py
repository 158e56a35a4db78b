"""matfac_amd -- MI355X-native (gfx950) drop-in for the ModelMF::train* hot path of
mohit-shrma/matfac.

The product is the C-ABI library ``libmfx.so`` (include/mfx.h, sources under
matfac_amd/csrc) and the C++ host classes under matfac_amd/host.  This Python
package is a thin ctypes view of that C ABI used by the tests and bench.py; it
contains no compute and no CPU fallback.
"""
from . import _lib  # noqa: F401
from .mfx import Ctx, MfxError, SgdOpts, EvalOut  # noqa: F401
from . import synth  # noqa: F401
