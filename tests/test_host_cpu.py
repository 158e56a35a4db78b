"""CPU-side tests of the host library (matfac_amd/host): the text-CSR loader that replaces gk_csr_Read /
gk_csr_CreateIndex, Data's nUsers/nItems rule (datastruct.cpp:23,91), the factor initialisation stream and
the synthetic generator.  The oracle's independent reader/writer is the checker."""
import ctypes as C

import numpy as np
import pytest

from matfac_amd import synth
from oracle import binding as orc


def host_read(path, want_cols=True):
    lib = synth._host()
    nr, nc, nz = C.c_int32(), C.c_int32(), C.c_int64()
    err = C.create_string_buffer(256)
    rc = lib.mfh_csr_read_text(path.encode(), C.byref(nr), C.byref(nc), C.byref(nz), None, None, None, None, None,
                               None, err, 256)
    if rc:
        raise IOError(err.value.decode())
    rp = np.empty(nr.value + 1, np.int64); ri = np.empty(nz.value, np.int32); rv = np.empty(nz.value, np.float32)
    cp = np.empty(nc.value + 1, np.int64); ci = np.empty(nz.value, np.int32); cv = np.empty(nz.value, np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.mfh_csr_read_text(path.encode(), C.byref(nr), C.byref(nc), C.byref(nz), P(rp), P(ri), P(rv),
                               P(cp) if want_cols else None, P(ci), P(cv), err, 256)
    assert rc == 0
    return nr.value, nc.value, rp, ri, rv, cp, ci, cv


def test_text_csr_loader_matches_oracle_reader(tmp_path):
    d = synth.make(dict(nU=150, nI=90, nnz=3000, K=0), seed=8)
    tr = d["train"]
    p = str(tmp_path / "t.csr")
    orc.write_csr_text(p, tr.nrows, tr.rowptr, tr.rowind, tr.rowval)
    nr, nc, rp, ri, rv, cp, ci, cv = host_read(p)
    onr, onc, orp, ori, orv = orc.read_csr_text(p)
    assert (nr, nc) == (onr, onc) == (tr.nrows, tr.ncols)
    assert np.array_equal(rp, orp) and np.array_equal(ri, ori) and np.array_equal(rv, orv)
    ocp, oci, ocv = orc.create_col_index(nr, nc, rp, ri, rv)
    assert np.array_equal(cp, ocp) and np.array_equal(ci, oci) and np.array_equal(cv, ocv)


def test_loader_edge_cases(tmp_path):
    p = str(tmp_path / "e.csr")
    open(p, "w").write("% header comment\n3 4.5 0 1\n\n\n7 2\n")      # unsorted row, two empty users, comment
    nr, nc, rp, ri, rv, cp, ci, cv = host_read(p)
    assert (nr, nc) == (4, 8)
    assert rp.tolist() == [0, 2, 2, 2, 3] and ri.tolist() == [3, 0, 7] and rv.tolist() == [4.5, 1.0, 2.0]
    assert cp.tolist() == [0, 1, 1, 1, 2, 2, 2, 2, 3] and ci.tolist() == [0, 0, 3]
    open(p, "w").write("0 1.0 2\n")                                       # odd token count
    with pytest.raises(IOError):
        host_read(p)
    with pytest.raises(IOError):
        host_read(str(tmp_path / "missing.csr"))
    open(p, "w").write("")                                                 # empty file: 0 x 0
    nr, nc, rp, *_ = host_read(p)
    assert (nr, nc) == (0, 0) and rp.tolist() == [0]


def test_data_shape_rule(tmp_path):
    """nUsers = train rows; nItems = 1 + max item over train, test AND val (datastruct.cpp:91)."""
    tr, te, va = (str(tmp_path / n) for n in ("tr", "te", "va"))
    open(tr, "w").write("0 5 1 3\n2 4\n")
    open(te, "w").write("6 1\n\n")
    open(va, "w").write("\n4 2\n")
    lib = synth._host()
    nU, nI, nz = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.mfh_data_shape(tr.encode(), te.encode(), va.encode(), C.byref(nU), C.byref(nI), C.byref(nz)) == 0
    assert (nU.value, nI.value, nz.value) == (2, 7, 3)


def test_init_factors_stream_and_generator_invariants():
    U, V = synth.init_factors(1, 6, 5, 8)
    Uo, Vo = orc.init_factors(1, 6, 5, 8)
    assert np.array_equal(U, Uo) and np.array_equal(V, Vo)
    _, V2 = synth.init_factors(1, 6, 5, 8, want_u=False)                   # V does not depend on U being requested
    assert np.array_equal(V2, Vo)
    a = synth.make("C1", seed=1)
    b = synth.make("C1", seed=1)
    c = synth.make("C1", seed=1, shard=1)
    f = a["full"]
    assert f.nnz == 100_000 and a["train"].nnz + a["val"].nnz + a["test"].nnz == f.nnz
    assert np.array_equal(f.rowind, b["full"].rowind) and np.array_equal(f.rowval, b["full"].rowval)   # deterministic
    assert not np.array_equal(f.rowind[:1000], c["full"].rowind[:1000])                                  # another user block
    assert np.diff(a["train"].rowptr).min() >= 1                            # every user keeps a train rating
    key = f.rowids().astype(np.int64) * f.ncols + f.rowind
    assert np.all(np.diff(key) > 0)                                         # sorted, no duplicate (user, item)
    assert set(np.unique(f.rowval)) <= set(np.arange(0.5, 5.01, 0.5).astype(np.float32))
    # same item catalogue across shards: the popular items coincide
    top = lambda m: set(np.argsort(np.bincount(m.rowind, minlength=1682))[-20:])
    assert len(top(f) & top(c["full"])) >= 12


def test_parallel_reader_number_formats_and_piece_cuts(tmp_path):
    """The mapped, multi-piece reader against the oracle's getline/strtol/strtof reader: every float spelling
    bit for bit, comment/empty lines and a missing final newline, on a file big enough for several pieces."""
    rng = np.random.default_rng(5)
    vals = rng.normal(3, 2, 400000).astype(np.float32)
    spell = ["%.9g", "%g", "%.3f", "%.17g", "%e", "%.1f", "%.12f"]
    special = ["inf", "-inf", "+3", ".5", "5.", "1e+5", "1E-3", "0x1.8p1", "1e-40", "3.4028235e38", "1e39", "0",
               "-0.0", "0.1", "16777217", "8388608.5", "8388609.5", "0.000000000000000000001", "123456789012345678",
               "1.00000005960464477539", "1.0000000596046447753906250", "1.00000017881393432617187500"]
    lines, k = [], 0
    for u in range(30000):
        if u % 997 == 0:
            lines.append("% a comment line that is not a user")
        n = int(rng.integers(0, 25))
        toks = []
        for _ in range(n):
            v = special[k % len(special)] if k % 53 == 0 else spell[k % len(spell)] % vals[k % len(vals)]
            toks.append("%d %s" % (int(rng.integers(0, 50000)), v))
            k += 1
        sep = "  " if u % 5 == 0 else " "
        lines.append(("\t" if u % 7 == 0 else "") + sep.join(toks) + (" \r" if u % 11 == 0 else ""))
    p = str(tmp_path / "big.csr")
    open(p, "w").write("\n".join(lines))                                   # no newline at the end
    assert len("\n".join(lines)) > 3 << 20
    nr, nc, rp, ri, rv, *_ = host_read(p, want_cols=False)
    onr, onc, orp, ori, orv = orc.read_csr_text(p)
    assert (nr, nc) == (onr, onc) and nr == 30000
    assert np.array_equal(rp, orp) and np.array_equal(ri, ori)
    assert np.array_equal(rv.view(np.uint32), orv.view(np.uint32))        # bit patterns: -0.0, inf, roundings
    open(p, "a").write("\n7 x\n")
    with pytest.raises(IOError) as e:
        host_read(p)
    assert "line %d" % (len(lines) + 1) in str(e.value)


def test_factor_files_text_and_lossless_binary(tmp_path):
    """writeMat keeps 6 significant digits (io.cpp:139-154, read back by the oracle's reader); the .binmat pair
    (io.cpp:172-184: one double per value) is lossless and is what readMat picks by extension."""
    lib = synth._host()
    rng = np.random.default_rng(2)
    M = (rng.normal(0, 1, (37, 11)) * 10.0 ** rng.integers(-6, 3, (37, 11))).astype(np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    t, b = str(tmp_path / "m.mat"), str(tmp_path / "m.binmat")
    assert lib.mfh_mat_write(t.encode(), P(M), 37, 11, 0) == 0 and lib.mfh_mat_write(b.encode(), P(M), 37, 11, 1) == 0
    back_t, back_b = np.empty_like(M), np.empty_like(M)
    assert lib.mfh_mat_read(t.encode(), P(back_t), 37, 11) == 0 and lib.mfh_mat_read(b.encode(), P(back_b), 37, 11) == 0
    assert np.array_equal(back_b.view(np.uint32), M.view(np.uint32))
    assert np.allclose(back_t, M, rtol=1e-5) and np.array_equal(back_t, orc.read_mat(t, 37, 11))
    assert np.array_equal(np.fromfile(b, np.float64).reshape(37, 11), M.astype(np.float64))
    assert lib.mfh_mat_read(b.encode(), P(back_b), 38, 11) != 0          # short file


def test_dropin_example_compiles_and_links_against_the_class_surface(tmp_path):
    """INTEGRATION.md section A as a program (examples/dropin_main.cpp): the reference's main() for --algo=mf against
    this repo's headers.  No GPU needed to compile and link; without a device it must fail loudly, not fall back."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dropin")
    cmd = ["g++", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "matfac_amd", "host"),
           os.path.join(root, "examples", "dropin_main.cpp"), "-L" + os.path.join(root, "matfac_amd"), "-lmfhost", "-lmfx",
           "-Wl,-rpath," + os.path.join(root, "matfac_amd"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([exe], capture_output=True, text=True).returncode != 0       # usage
    from matfac_amd import mfx
    try:
        mfx.Ctx(0).close()
        has_gpu = True
    except mfx.MfxError:
        has_gpu = False
    if not has_gpu:
        files = []
        for name in ("tr", "te", "va"):
            p = str(tmp_path / name)
            open(p, "w").write("0 4.0 1 3.0\n1 2.0\n")
            files.append(p)
        run = subprocess.run([exe] + files + [str(tmp_path / "run"), "sgd", "2", "1"], capture_output=True, text=True, timeout=120)
        assert run.returncode != 0 and "mfx_create failed" in run.stderr            # no CPU fallback


def test_train_test_val_splitter_files(tmp_path):
    """writeTrainTestValMat (io.cpp:410-459): the three files are the colour classes of the oracle's restatement of the
    mt19937 colouring (test drawn WITH replacement, val exactly valPc*nnz), each with the full shape and the row order."""
    d = synth.make(dict(nU=120, nI=70, nnz=2500, K=0), seed=4)
    m = d["full"]
    lib = synth._host()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    files = [str(tmp_path / n) for n in ("train.csr", "test.csr", "val.csr")]
    assert lib.mfh_write_train_test_val(C.c_int32(m.nrows), C.c_int32(m.ncols), P(m.rowptr), P(m.rowind), P(m.rowval),
                                        files[0].encode(), files[1].encode(), files[2].encode(), C.c_float(0.1), C.c_float(0.2),
                                        C.c_int32(7)) == 0
    color = orc.split_colors(m.nnz, 0.1, 0.2, 7)
    assert (color == 2).sum() == int(np.float32(0.2) * m.nnz) and 0 < (color == 1).sum() <= int(np.float32(0.1) * m.nnz)
    rows = m.rowids()
    for c, f in enumerate(files):
        assert sum(1 for _ in open(f)) == m.nrows                  # one line per user, empty rows included
        nr, nc, rp, ri, rv, *_ = host_read(f, want_cols=False)
        sel = color == c
        assert nr == m.nrows
        assert np.array_equal(ri, m.rowind[sel]) and np.array_equal(rv, m.rowval[sel])      # ratings are multiples of 0.5: "%f" is exact
        assert np.array_equal(np.diff(rp), np.bincount(rows[sel], minlength=m.nrows))
    txt = open(files[0]).readline()
    assert txt.startswith(" ") and "." in txt                      # GKlib's " %d %f" entries


def test_rand_mat_csr_writer(tmp_path):
    """writeRandMatCSR (io.cpp:726-787): the sampled pairs are the oracle's restatement of the mt19937 sequence; every user
    and every item occurs; the rating is the double dot product of the given factors printed with ostream precision."""
    nU, nI, K, nnz = 60, 40, 5, 900
    rng = np.random.default_rng(3)
    U = rng.normal(0, 1, (nU, K)); V = rng.normal(0, 1, (nI, K))
    lib = synth._host()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    f = str(tmp_path / "rand.csr")
    assert lib.mfh_write_rand_mat_csr(f.encode(), P(U), P(V), nU, nI, K, 11, nnz) == 0
    pairs = orc.rand_pairs(nU, nI, 11, nnz)
    assert len(pairs) >= nnz
    nr, nc, rp, ri, rv, *_ = host_read(f, want_cols=False)
    assert nr == nU and rp[-1] == len(pairs)
    assert np.array_equal(np.repeat(np.arange(nU), np.diff(rp)), pairs[:, 0]) and np.array_equal(ri, pairs[:, 1])
    assert np.all(np.diff(rp) >= 1) and np.unique(ri).size == nI
    exact = np.einsum("ij,ij->i", U[pairs[:, 0]], V[pairs[:, 1]])
    assert np.allclose(rv, exact, rtol=2e-5, atol=1e-6)            # 6 significant digits in the file


def test_host_classes_report_a_missing_device_instead_of_exiting():
    """The library classes throw MfxError (caught at the C entry points) where round 1 called exit(-2): without a GPU
    ModelMF::train must come back with MFX_E_NODEVICE and leave the embedding process alive."""
    from tests.conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    d = synth.make(dict(nU=40, nI=30, nnz=400, K=0), seed=1)
    tr, va, te = d["train"], d["val"], d["test"]
    lib = synth._host()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    K = 4
    bufs = [np.zeros((d["nUsers"], K), np.float32), np.zeros((d["nItems"], K), np.float32)]
    rc = lib.mfh_train(b"sgd", C.c_int32(tr.nrows), P(tr.rowptr), P(tr.rowind), P(tr.rowval), C.c_int32(tr.ncols), P(va.rowptr), P(va.rowind),
                       P(va.rowval), C.c_int32(va.ncols), P(te.rowptr), P(te.rowind), P(te.rowval), C.c_int32(te.ncols), C.c_int32(K), C.c_int32(2),
                       C.c_int32(1), C.c_float(0.01), C.c_float(0.01), C.c_float(0.01), None, P(bufs[0]), P(bufs[1]), None, None, None, None, None)
    assert rc == -6                      # MFX_E_NODEVICE



@pytest.mark.parametrize("n", [0, 1, 2, 999, 65535, 65536, 65537, 70001, 300007, 1048576, 2000003])
def test_epoch_shuffle_is_std_shuffle_bit_for_bit(n):
    """ModelMF::train's per-epoch std::shuffle of the index list (modelMF.cpp:76-81) is the host's largest cost next to the replay
    on the GPU; mfhShuffle draws the swap positions a block ahead (same distribution object, same calls; for long lists on a
    second thread, from a restated mt19937 + Lemire multiply-shift that a self-check holds to the library's) and requests their
    cache lines before the swaps follow.  Same permutation and same mt19937 state afterwards as the library call, on both of
    libstdc++'s paths (two positions per draw up to 65 536 entries, the plain loop beyond)."""
    lib = synth._host()
    lib.mfh_shuffle_check.argtypes = [C.c_int64, C.c_uint32, C.POINTER(C.c_double)]
    lib.mfh_shuffle_check32.argtypes = [C.c_int64, C.c_uint32, C.POINTER(C.c_double)]
    for seed in (1, 12345):
        secs = (C.c_double * 3)()
        assert lib.mfh_shuffle_check(n, seed, secs) == 1
        assert lib.mfh_shuffle_check32(n, seed, secs) == 1          # a list of 32-bit entries: the same swaps, the same state
    # this image's libstdc++ (GCC 11) is the one the restated generator + distribution were written against: from 2^20 entries on a
    # second thread swaps while the first draws with them (form 2); another library would fail the self-check and report 0 or 1
    assert secs[2] == 2.0


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 12345, 3_000_017])
def test_positions_of_the_epoch_shuffle_are_std_shuffles(n):
    """Round 4: for long lists the host draws only the POSITIONS of std::shuffle (mfhShufflePositions: the generator's stream on two
    threads) and the device applies the swaps (mfx_sgd_apply_swaps32, tests/test_setup_gpu.py).  Applied one by one on the host they
    must give std::shuffle's list and leave the generator in std::shuffle's state; below 2^20 entries the form does not apply."""
    lib = synth._host()
    lib.mfh_shuffle_positions_check.argtypes = [C.c_int64, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]
    for seed in (1, 4242):
        secs = (C.c_double * 1)()
        pos = np.empty(n, np.uint32)
        assert lib.mfh_shuffle_positions_check(n, seed, pos.ctypes.data_as(C.c_void_p), secs) == 1
        assert pos[0] == 0 and np.all(pos <= np.arange(n, dtype=np.uint32))
    assert lib.mfh_shuffle_positions_check(70001, 1, None, None) == 2
