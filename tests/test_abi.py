"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mfx.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from matfac_amd import _lib, mfx
from tests.conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfx_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ["mfx_create", "mfx_destroy", "mfx_set_csr", "mfx_set_model", "mfx_set_factors", "mfx_get_factors",
                 "mfx_compute_invalid", "mfx_sgd_epoch", "mfx_eval", "mfx_eval2", "mfx_als_half_sweep", "mfx_ccdpp_rank1",
                 "mfx_snapshot_best", "mfx_allreduce_item_factors", "mfx_prof_get"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, "libmfx.so lacks: %s" % missing


def test_header_is_plain_c():
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "t.c")
        open(p, "w").write('#include "mfx.h"\nint main(void){mfx_sgd_opts o; (void)o; return MFX_OK;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", p,
                               "-o", os.path.join(td, "t.o")])


def test_struct_layouts_match_ctypes():
    """sizeof / offsetof of the two structs of include/mfx.h as gcc lays them out == the ctypes mirrors of matfac_amd/mfx.py"""
    import subprocess, tempfile
    fields = [f for f, _ in mfx.SgdOpts._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "mfx.h"\nint main(void){printf("%zu %zu", sizeof(mfx_sgd_opts), sizeof(mfx_eval_out));\n'
    prog += "".join('printf(" %%zu", offsetof(mfx_sgd_opts, %s));\n' % f for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(td, "t.c"), "-o", os.path.join(td, "t")])
        got = [int(x) for x in subprocess.check_output([os.path.join(td, "t")]).split()]
    assert got[0] == C.sizeof(mfx.SgdOpts) == 64
    assert got[1] == C.sizeof(mfx.EvalOut) == 32
    assert got[2:] == [getattr(mfx.SgdOpts, f).offset for f in fields]


def test_no_cpu_fallback_without_a_device():
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(mfx.MfxError) as e:
        mfx.Ctx(0)
    assert e.value.code == mfx.E_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """Nothing under matfac_amd/ or include/ may import, link or call the oracle."""
    bad = []
    for base in ("matfac_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"liboracle|oracle\.h|from oracle|import oracle|orc_[a-z]", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
