"""Host logic of the CCD++ passes: the block plan of the padded views (matfac_amd/csrc/ccd_blocks.h).  tests/native/blocks_check.hip is
compiled with hipcc and run on the CPU (no kernel is launched): every trip of a region is handled by exactly one live record, a
group's records past its last trip are dead, every slot is written once, a piece's slots are consecutive and cover exactly its
lanes in order, the workgroups carry equal numbers of trips, the quad-interleaved residual order is a bijection.  Below that: checks
of the generated ISA (wait states inside inline assembly, landing registers of the tagged replay)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "blocks_check.hip")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("blocks") / "blocks_check")
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-w", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "matfac_amd", "csrc"), SRC, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def _pieces(rng, n, max_len, r0):
    """n pieces in memory order from r0 on: lengths 1..max_len (a quarter of them of one or two entries), each padded to 8."""
    lens = rng.integers(1, max_len + 1, n)
    short = rng.random(n) < 0.25
    lens[short] = rng.integers(1, 3, short.sum())
    plen = (lens + 7) // 8 * 8
    b = r0 + np.concatenate([[0], np.cumsum(plen)[:-1]])
    return np.stack([b, b + plen], 1)


@pytest.mark.parametrize("n,max_len,nwg,r0", [(1, 5, 1, 0), (7, 5000, 1, 128), (3000, 1024, 5, 0), (20000, 300, 16, 128 * 77), (5000, 40, 3, 0),
                                              (40000, 60, 512, 1280), (3, 100000, 7, 0)])
def test_block_plan_covers_every_piece_once(checker, n, max_len, nwg, r0):
    rng = np.random.default_rng(n + max_len)
    pc = _pieces(rng, n, max_len, r0)
    r1 = (int(pc[-1, 1]) + 127) // 128 * 128 + 128 * int(rng.integers(0, 3))        # a tail no piece owns, sometimes whole trips of it
    text = "".join("%d %d\n" % (b, e) for b, e in pc)
    r = subprocess.run([checker, str(r0), str(r1), str(nwg)], input=text, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    parts = r.stdout.split()
    wmin, wmax = int(parts[parts.index("min") + 1]), int(parts[parts.index("max") + 1])
    assert wmax - wmin <= 64                           # chunks of 64 trips dealt round-robin: within one chunk of each other


def test_block_plan_refuses_what_the_loop_cannot_run(checker):
    for text, r0, r1 in [("4 16\n", 0, 128),            # a piece that does not start on a multiple of 8
                         ("0 16\n8 24\n", 0, 128),      # overlapping pieces
                         ("0 136\n", 0, 128),           # a piece behind the region
                         ("0 16\n", 0, 130)]:           # a region that is not whole trips
        r = subprocess.run([checker, str(r0), str(r1), "2"], input=text, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout.startswith("REFUSED"), (text, r.stdout)


_ASM = {}


def _device_asm(src, tmp_path):
    if src in _ASM:
        return _ASM[src]
    _ASM[src] = _compile_asm(src, tmp_path)
    return _ASM[src]


def _compile_asm(src, tmp_path):
    out = str(tmp_path / (os.path.basename(src) + ".s"))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-w", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "matfac_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


@pytest.mark.parametrize("src,nmin", [("ccd.hip", 10), ("ccd_cols.hip", 10), ("sgd_flow.hip", 1)])
def test_wide_stores_written_as_inline_assembly_carry_their_wait_state(src, nmin, tmp_path):
    """On gfx940 and later a store of more than 64 bits needs TWO wait states before a VALU may overwrite its data registers, and
    the compiler's hazard recognizer does not look inside an asm statement: behind the last 16-byte store between ;;#ASMSTART and
    ;;#ASMEND there must be an `s_nop 1` (or more) INSIDE the statement (mfx_blk_store / mfx_blk_store_res in ccd_blocks.h, fl_store
    in sgd_flow.hip; a store followed by another store of the same statement has that one in between).
    Round 4's first block loop carried `s_nop 0`, which was enough before gfx940: 41 of 1 500 rows came out up to 13 ulp off
    (tests/test_ccd_gpu.py) -- the pairs were stored with low words the next instructions had already overwritten."""
    import re
    text = _device_asm(os.path.join(ROOT, "matfac_amd", "csrc", src), tmp_path)
    blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, flags=re.S)
    wide = [b for b in blocks if "store_dwordx4" in b]
    assert len(wide) >= nmin, "%d inline-assembly 16-byte stores found in %s" % (len(wide), src)
    for b in wide:
        lines = [l.strip() for l in b.strip().split("\n") if l.strip()]
        k = [i for i, l in enumerate(lines) if "store_dwordx4" in l.split()[0]]
        last = k[-1]
        assert last + 1 < len(lines) and re.match(r"s_nop ([1-9]|1[0-5])$", lines[last + 1]), b
        assert all(i + 1 in k or i == last for i in k), b          # stores of one statement follow each other directly


def test_permlane_swaps_written_as_inline_assembly_carry_their_wait_states(tmp_path):
    """gfx950 wants two wait states between a VALU write of a VGPR and a v_permlane32_swap that reads it (and before a VALU reads
    what the swap wrote); the hazard recognizer does not look inside an asm statement and the compiler puts the "+v" copies right
    in front of it.  Every swap between ;;#ASMSTART and ;;#ASMEND must sit between two s_nop 1 of its own (MFX_SWAP32, als.hip):
    a round-3 rebuild produced `v_mov v145, v62` directly followed by the bare swap in als_reduce_kernel -- rows with several
    segments came out 5 % off (scripts/als_reduce_check.py)."""
    import re
    text = _device_asm(os.path.join(ROOT, "matfac_amd", "csrc", "als.hip"), tmp_path)
    blocks = [b for b in re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, flags=re.S) if "v_permlane32_swap" in b]
    assert len(blocks) >= 33
    for b in blocks:
        lines = [l.strip() for l in b.strip().split("\n") if l.strip()]
        k = [i for i, l in enumerate(lines) if l.startswith("v_permlane32_swap")]
        assert all(i >= 1 and lines[i - 1] == "s_nop 1" and i + 1 < len(lines) and lines[i + 1] == "s_nop 1" for i in k), b
    # and no swap outside an asm statement (the builtin is not used: the compiler merged most of the 33 calls into three)
    outside = re.sub(r";;#ASMSTART\n.*?;;#ASMEND", "", text, flags=re.S)
    assert "v_permlane32_swap" not in outside


def test_landing_registers_of_the_tagged_replay_are_never_copied():
    """sgd_flow_tag_kernel / sgd_flow_wide_kernel request the rows of the next queue positions with inline-assembly loads and
    wait for them with counted s_waitcnt statements the compiler does not understand: for it a landing register holds its value
    from the load statement on, so any v_mov it schedules between the load and the wait copies a register whose data has not
    arrived (the first GPU run of the wide kernel computed wrong rows that way).  Checked on the compiled code:
      * 16-lane kernel: the loads of all steps of one pipeline slot target the same registers -- exactly LA x (loads per request)
        destinations that are loaded five times or more, plus the few scratch destinations of the probe loop;
      * wide kernel, generic steps: every move that reads or writes a register an `sc1` row load lands in sits behind an
        `s_waitcnt vmcnt(0)` with no such load in between (round 4: with the pole path in the kernel the allocator parks the four slots
        in other registers across a pole block -- moves at block boundaries, where nothing is in flight, are what is allowed);
      * wide kernel, pole blocks: the landing registers are accumulation registers named in the asm text (a0 .. a31), which the compiler
        never allocates: no accumulation register appears outside an asm statement, the loads into them are LAP x C distinct pairs,
        and nothing spills."""
    import re
    import tempfile
    import pathlib
    with tempfile.TemporaryDirectory() as d:
        text = _device_asm(os.path.join(ROOT, "matfac_amd", "csrc", "sgd_flow.hip"), pathlib.Path(d))
    lines = text.split("\n")
    seen = 0

    def regs(tok):                                # "v[14:15]" / "v7" -> set of register numbers
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    for i, l in enumerate(lines):
        m = re.match(r"^_ZN12_GLOBAL__N_1\d+sgd_flow_(tag|wide)_kernelILi(\d+)ELi(\d+)E(?:Li(\d+)E)?", l)
        if not m or ":" not in l:                 # the label line of the kernel ("name:   ; @name")
            continue
        j = i
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        if m.group(1) == "tag":
            L, C = int(m.group(2)), int(m.group(3))
            la, per = (4 if C <= 2 else 2), 2 * C
            dst = {}
            for b in body:
                k = re.match(r"\s*buffer_load_dwordx[24] (v\[\d+:\d+\]), v\d+, s\[\d+:\d+\], 0 offen sc1", b)
                if k:
                    dst[k.group(1)] = dst.get(k.group(1), 0) + 1
            landing = [r for r, n in dst.items() if n >= 5]
            assert la * per <= len(landing) <= la * per + 2 and len(dst) <= la * per + 6, (l[:60], sorted(dst.items()))
        else:
            C = int(m.group(2))
            lap = 16 if C == 1 else 8 if C == 2 else 4
            land = set()
            for b in body:
                k = re.match(r"\s*buffer_load_dwordx2 (v\[\d+:\d+\]), v\d+, s\[\d+:\d+\], 0 offen sc1", b)
                if k:
                    land |= regs(k.group(1))
            for n, b in enumerate(body):
                k = re.match(r"\s*v_mov_b(?:32|64)(?:_e32|_e64)? (v\[\d+:\d+\]|v\d+), (v\[\d+:\d+\]|v\d+)\s*$", b)
                if not k or not ((regs(k.group(1)) | regs(k.group(2))) & land):
                    continue
                touched = regs(k.group(1)) | regs(k.group(2))
                ok = None
                for t in range(n - 1, -1, -1):
                    x = body[t].strip()
                    if x.startswith("s_waitcnt vmcnt(0)"):
                        ok = True
                        break
                    q = re.match(r"buffer_load_dwordx2 (v\[\d+:\d+\]), v\d+, s\[\d+:\d+\], 0 offen sc1", x)
                    if q and regs(q.group(1)) & touched:
                        ok = False
                        break
                assert ok is not False, (l[:70], n, b.strip())      # (None: the kernel's first instructions, nothing requested yet)
            # the pole path: accumulation registers only inside asm statements, LAP x C landing pairs, no scratch
            joined = "\n".join(body)
            outside = re.sub(r";;#ASMSTART\n.*?;;#ASMEND", "", joined, flags=re.S)
            # (the compiler may park values of its own in accumulation registers ABOVE the landing ones: a32.. in the hybrid C = 4 kernel)
            nland = 2 * lap * C
            mine = [x for x in outside.split("\n") if not x.strip().startswith(";")
                    and any(int(a) < nland for a in re.findall(r"\ba\[?(\d+)", x))]
            assert not mine, (l[:70], mine[:4])
            pairs = set(re.findall(r"buffer_load_dwordx2 (a\[\d+:\d+\])", joined))
            assert len(pairs) == lap * C, (l[:70], sorted(pairs))
            assert "scratch_" not in joined
        seen += 1
    assert seen >= 30          # 7 rank shapes x 3 arithmetic modes of the 16-lane kernel, 4 x 3 of the wide one
