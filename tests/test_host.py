"""CPU-only tests of the product's host side: C-ABI surface, index construction and reading, the
suffix sorter, the read pipeline, argument checking, and the shared kernel core (gm_core.hpp)
compiled for the host.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from oracle import pyoracle as po
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gpumap.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpumap_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = C.CDLL(ma.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), "libgpumap.so does not export %s" % name
    assert declared == set(ma.API), "python binding and header disagree: %s" % (declared ^ set(ma.API))
    assert b"gfx950" in ma.lib().gpumap_version()


def test_header_is_c99_and_the_integration_sketch_compiles_against_the_reference():
    """include/gpumap.h as a maintainer of the reference would see it: plain C99 (the reference builds with
    -std=c99, Makefile:16), no HIP or C++ header behind it; and, where the reference's headers are present (this
    container, not the GPU box), the binding of INTEGRATION.md section 1 (tools/integration/mapper_gpumap.c) compiles
    against them: every reference function it calls exists with the signature it assumes."""
    import shutil, subprocess, tempfile
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    with tempfile.NamedTemporaryFile("w", suffix=".c") as f:
        f.write('#include "gpumap.h"\nint main(void) { gpumap_hits h; (void)h; return gpumap_version() == 0; }\n')
        f.flush()
        r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # the structs the python binding mirrors have the size and the field offsets the C header gives them (a field added
    # to one side only makes the library write past the other's allocation)
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "sz.c"), os.path.join(d, "sz")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "gpumap.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                             'sizeof(gpumap_hits), offsetof(gpumap_hits, loc_key), offsetof(gpumap_hits, line_bytes), sizeof(gpumap_index_view), '
                             'sizeof(gpumap_devindex)); return 0; }\n')
        r = subprocess.run([gcc, "-std=c99", "-I", inc, src, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(ma._Hits), ma._Hits.loc_key.offset, ma._Hits.line_bytes.offset, C.sizeof(ma.IndexView), C.sizeof(ma.DevIndex)]
    ref = "/root/reference/src"
    if not os.path.isfile(os.path.join(ref, "mapper.h")):
        pytest.skip("the reference's headers are not on this machine")
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", ref, "-I", inc,
                        os.path.join(ROOT, "tools", "integration", "mapper_gpumap.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_contains_gfx950_code_object():
    blob = open(ma.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob and b"k_search" in blob


def test_null_arguments_follow_error_convention(capfd):
    L = ma.lib()
    assert not L.gpumap_index_read(None)
    assert not L.gpumap_index_build(None, b"x")
    assert not L.gpumap_create(None, 0)
    assert L.gpumap_index_view_of(None, None) == -1
    assert L.gpumap_occ_all(None, 1, None, None) == -1
    assert L.gpumap_map_batch(None, None, None, 0, 0, 0, None) == -1
    assert L.gpumap_batch_run(None, None, 0, 0) == -1
    assert L.gpumap_mapper(None, None, None, 0, None) == -1
    err = capfd.readouterr().err
    # func \t (file:line) error: msg   (src/errhandler.c:12-15)
    assert re.search(r"gpumap_index_read\t\(.*gm_index\.cpp:\d+\) error: argument 'filename_base' is NULL\.", err)
    assert not L.gpumap_index_read(b"/nonexistent/base")


def test_index_read_roundtrip_and_view(workdir):
    base = common.build_golden_index(workdir, "small")
    ix = ma.HostIndex.read(base)
    orc = po.Index(base)
    v = ix.view
    assert v.txt_len == orc.txt_len and v.sar_bits == orc.sar_bits and v.seq_cnt == 2 and v.rc_flag == 1
    assert v.wil_cnt == 4 and v.occ_mark_intv == 14 and v.occ_word_size == 64 and v.sym_count == 5
    assert ix.seq_names() == ["seqA", "seqB"]                # name stops at the first space (src/index.c:295-297)
    assert [int(C.cast(v.c, C.POINTER(C.c_uint64))[i]) for i in range(6)] == orc.c
    n = v.txt_len
    text = np.ctypeslib.as_array(C.cast(v.text, C.POINTER(C.c_uint8)), shape=(n,))
    comp = np.array([3, 2, 1, 0, 4, 5], dtype=np.uint8)
    # F $ revcomp(F) $  (src/index_txt.c:314-334)
    assert text[n // 2 - 1] == 5 and text[n - 1] == 5
    assert np.array_equal(text[n // 2:n - 1], comp[text[:n // 2 - 1][::-1]])
    for p in (0, 5, n // 2 - 2, n // 2, n - 2):
        assert ix.pos_to_str(p) == orc.pos_to_str(p)
    with pytest.raises(ma.GpumapError):
        ix.pos_to_str(n)


def test_fasta_quirks_match_reference_rules(workdir):
    fa = os.path.join(workdir, "quirks.fa")
    with open(fa, "w") as f:
        f.write(">  s1   desc\nACGTxacgt\n\nNNAC\n>s2\nTTTT")      # no trailing newline, blank line, junk letters
    ix = ma.HostIndex.build(fa, os.path.join(workdir, "quirks"))
    assert ix.seq_names() == ["s1", "s2"]
    v = ix.view
    text = np.ctypeslib.as_array(C.cast(v.text, C.POINTER(C.c_uint8)), shape=(v.txt_len,))
    assert text[:v.txt_len // 2].tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 4, 0, 1, 5, 3, 3, 3, 3, 5]
    # duplicate names and a file that does not start with '>' are errors
    with open(fa, "w") as f:
        f.write(">a\nAC\n>a\nGT\n")
    with pytest.raises(ma.GpumapError):
        ma.HostIndex.build(fa, os.path.join(workdir, "dup"))
    with open(fa, "w") as f:
        f.write("ACGT\n")
    with pytest.raises(ma.GpumapError):
        ma.HostIndex.build(fa, os.path.join(workdir, "nohdr"))


def _sa(text):
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3, "$": 5}.get(c.upper(), 4) for c in text], dtype=np.uint8)
    out = np.empty(len(codes), dtype=np.int64)
    assert ma.lib().gpumap_suffix_sort(codes.ctypes.data_as(C.c_void_p), len(codes), out.ctypes.data_as(C.c_void_p)) == 0
    return out.tolist()


def test_suffix_sorter_known_answers():
    # test/test_index_sar.c:8-17 (texts with N, junk letters and several separators)
    assert _sa("TAGCNTCGACA$") == [11, 10, 8, 1, 9, 6, 3, 7, 2, 0, 5, 4]
    assert _sa("TGNGA$TCGAT$") == [11, 5, 4, 9, 7, 3, 8, 1, 10, 6, 0, 2]
    assert _sa("AAaaAAaaA$") == [9, 8, 7, 6, 5, 4, 3, 2, 1, 0]
    assert _sa("acgTPGaKcnN$") == [11, 0, 6, 1, 8, 5, 2, 3, 10, 7, 4, 9]
    assert _sa("$TgAgcatGGC$acTCGA$") == [18, 11, 0, 17, 12, 3, 6, 10, 5, 15, 13, 16, 2, 9, 4, 8, 14, 1, 7]
    # :105-126 a 10 000-symbol periodic text
    sa = _sa("TGCAAAAAA$" * 1000)
    for k in range(10):
        assert sa[1000 * k:1000 * (k + 1)] == [9999 - k - 10 * i for i in range(1000)]


def test_suffix_sorter_random_against_naive():
    rng = np.random.default_rng(3)
    for n in (1, 2, 50, 700):
        t = "".join(rng.choice(list("ACGTN$AC"), n))
        naive = sorted(range(n), key=lambda i: [("$ACGTN".index(c)) for c in t[i:]])
        assert _sa(t) == naive


def test_read_pipeline_matches_reference_io_tests(workdir):
    """test/test_io.c:70-190 on the reference's own example files."""
    def dump(name, buf):
        out = os.path.join(workdir, name + ".dump")
        n = ma.lib().gpumap_reads_dump(os.path.join(common.GOLDEN, name).encode(), buf, out.encode())
        return n, [c.strip("\n").split("\n") for c in open(out).read().split("#chunk\n")]
    n, chunks = dump("io_input.raw", 1 << 20)
    assert n == 6 and len(chunks) == 1
    assert chunks[0][3:] == ["4\tACGTACATGTATGACAC", "5\tNNNNNNNNACGTACGCC", "6\tNNNNNNLLYYYYJDFLS"]
    n, chunks = dump("io_input.fasta", 80)
    assert n == 6 and [len(c) for c in chunks] == [4, 2]
    assert chunks[0][0] == "seq1\tATGCGTACGTCGTATCA" and chunks[1] == ["seq5\tNNNNNNNNACGTACGCC", "seq6\tNNNNNNLLYYYYJDFLS"]
    n, chunks = dump("io_input.fastq", 50)
    assert n == 6 and [len(c) for c in chunks] == [2, 2, 2]
    assert chunks[0] == ["seq1\tATGCGTACGTCGTATCA", "seq2\tGTATCGACTACGAGCTA"]
    # inconsistent quality length is an error (src/seqread.c:47-49)
    bad = os.path.join(workdir, "bad.fastq")
    open(bad, "w").write("@r\nACGT\n+\nIII\n")
    assert ma.lib().gpumap_reads_dump(bad.encode(), 1 << 20, (bad + ".dump").encode()) == -1
    assert ma.lib().gpumap_reads_dump(b"/nonexistent", 10, b"/tmp/x") == -1


def _reference_reader_model(data, max_bytes):
    """io_stream_read_seq (src/io.c:113-227) + seqread_parse_* (src/seqread.c:14-164) line by line: getline,
    C-string content (stops at a NUL), chunk ends at the first record boundary with max_bytes behind it.
    Returns the chunks as lists of (tag, letters), or None where the reference fails (quality length)."""
    lines, pos = [], 0
    while pos < len(data):
        e = data.find(b"\n", pos)
        e = len(data) if e < 0 else e + 1
        lines.append(data[pos:e])
        pos = e
    P = 4 if data[:1] == b"@" else (2 if data[:1] == b">" else 1)

    def content(b):
        b = b[:-1] if b.endswith(b"\n") else b
        z = b.find(b"\0")
        return b if z < 0 else b[:z]
    chunks, i, lineno = [], 0, 1
    while True:
        cur, cnt, k, tag, seq = [], 0, 0, b"", b""
        while i < len(lines):
            c = content(lines[i])
            f = k % P
            if P == 1:
                cur.append((str(lineno).encode(), c))
            elif f == 0:
                tag = c[1:] if len(c) > 0 else c
            elif f == 1:
                seq = c
                if P == 2:
                    cur.append((tag, seq))
            elif f == 3:
                if len(c) != len(seq):
                    return None
                cur.append((tag, seq))
            cnt += len(lines[i]); i += 1; k += 1; lineno += 1
            if cnt >= max_bytes and k % P == 0:
                break
        if not cur:
            return chunks
        chunks.append(cur)
        if i >= len(lines) and not (cnt >= max_bytes and k % P == 0):
            return chunks


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_block_reader_against_line_by_line_model(workdir, threads, monkeypatch):
    """Random files (ragged lines, empty lines, NULs, CRs, no newline at the end, truncated last record) cut
    at random chunk sizes: the block reader with its threads gives what the reference's getline loop gives."""
    monkeypatch.setenv("GPUMAP_HOST_THREADS", threads)
    rng = np.random.default_rng(int(threads))
    alphabet = np.frombuffer(b"ACGTNacgtn\r\0xy", dtype=np.uint8)

    def rnd_line(lo, hi):
        return alphabet[rng.integers(0, len(alphabet), int(rng.integers(lo, hi)))].tobytes()
    for case in range(63):
        fmt = ("fastq", "fasta", "raw")[case % 3]
        recs = []
        big = case >= 60                                  # large enough for the slices of every thread to hold lines
        for r in range(40000 if big else int(rng.integers(0, 400))):
            s = rnd_line(0, 120)
            if fmt == "fastq":
                q = b"I" * len(s.split(b"\0")[0])
                recs += [b"@t%d x" % r, s, b"+", q]
            elif fmt == "fasta":
                recs += [b">t%d" % r, s]
            else:
                recs += [s if not s.startswith((b"@", b">")) or r else b"A" + s]
        if fmt == "raw" and recs and recs[0][:1] in (b"@", b">"):
            recs[0] = b"A" + recs[0]
        if recs and rng.random() < 0.3:
            recs = recs[:len(recs) - int(rng.integers(0, 3))]          # last record incomplete
        data = b"\n".join(recs) + (b"\n" if recs and rng.random() < 0.7 else b"")
        path = os.path.join(workdir, "rnd_reader_%s_%d" % (threads, case))
        open(path, "wb").write(data)
        for max_bytes in ((1 << 20, 3_000_000, 1 << 24) if big else
                          (1, int(rng.integers(2, 300)), int(rng.integers(300, 5000)), 1 << 22)):
            want = _reference_reader_model(data, max_bytes)
            n = ma.lib().gpumap_reads_dump(path.encode(), max_bytes, (path + ".dump").encode())
            if want is None:
                assert n == -1
                continue
            got = open(path + ".dump", "rb").read()
            exp = b"#chunk\n".join(b"".join(t + b"\t" + q + b"\n" for t, q in c) for c in want)
            assert n == sum(len(c) for c in want), (case, fmt, max_bytes)
            assert got == exp, (case, fmt, max_bytes)
            # the same cut, blocks left unparsed for the device (gm_reads_next_block): every block parsed by the model's
            # rules gives the chunk's reads, and its description says what it holds
            n2 = ma.lib().gpumap_reads_dump_blocks(path.encode(), max_bytes, (path + ".blocks").encode())
            assert n2 == n, (case, fmt, max_bytes)
            blob, at, lineno = open(path + ".blocks", "rb").read(), 0, 1
            for c in want:
                e = blob.index(b"\n", at)
                head = blob[at:e].split()
                assert head[0] == b"#block"
                n_rec, nbytes, per, n_lines, first = (int(x) for x in head[1:])
                block = blob[e + 1:e + 1 + nbytes]
                at = e + 1 + nbytes
                assert (n_rec, first) == (len(c), lineno), (case, fmt, max_bytes)
                lines = block.split(b"\n")
                if lines and lines[-1] == b"":
                    lines.pop()
                assert len(lines) == n_lines and per == {"fastq": 4, "fasta": 2, "raw": 1}[fmt] and n_rec == n_lines // per
                lineno += n_lines

                def content(b):
                    z = b.find(b"\0")
                    return b if z < 0 else b[:z]
                for r, (tag, seq) in enumerate(c):
                    assert content(lines[per * r + (0 if per == 1 else 1)]) == seq
                    if per > 1:
                        assert content(lines[per * r])[1:] == tag
                    else:
                        assert str(first + r).encode() == tag
            assert at == len(blob)


# ---------------------------------------------------------------- shared kernel core on the host

@pytest.fixture(scope="module")
def emu():
    path = os.path.join(ROOT, "tests", "hostemu", "_build", "libhostemu.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostemu")])
    E = C.CDLL(path)
    E.emu_create.restype = C.c_void_p
    E.emu_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    E.emu_search.restype = C.c_int64
    E.emu_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_int64))]
    E.emu_search_packed.restype = C.c_int64
    E.emu_search_packed.argtypes = E.emu_search.argtypes
    E.emu_occ_all.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    E.emu_free.argtypes = [C.c_void_p]
    E.emu_destroy.argtypes = [C.c_void_p]
    return E


def test_kernel_core_block_layout_and_schedule_on_host(emu, workdir):
    """gm_core.hpp (rank blocks, node packing, phase table, per-node step), run sequentially on the
    host, gives the oracle's ranks, hits and extension-step counts: both in the plain form
    (make_child over all five symbols) and in the form k_search runs (one child per pass chosen by
    valid_children / child_meta, continuations, phase records)."""
    base = common.build_golden_index(workdir, "small")
    hix = ma.HostIndex.read(base)
    v = hix.view
    h = emu.emu_create(v.occ, v.occ_mark_intv, v.c, v.txt_len)
    orc = po.Index(base)
    rng = np.random.default_rng(5)
    for p in [-1, 0, 126, 127, 128, v.txt_len - 1] + rng.integers(-1, v.txt_len, 3000).tolist():
        out = (C.c_int64 * 5)()
        emu.emu_occ_all(h, int(p), out)
        assert list(out) == orc.occ_all(int(p))
    reads = common.load_reads("small.reads.txt")
    for tau in (0, 1, 2, 4, 7):
        for r in reads[::7] + reads[-20:]:
            if not r or len(r) > 256:
                continue
            q = orc.encode(r)
            hp = C.POINTER(po.OrcHit)()
            ne = C.c_int64()
            nref = po.lib().orc_blocksearch(orc._h, q.ctypes.data_as(C.POINTER(C.c_uint8)), len(q), tau, C.byref(hp), C.byref(ne))
            want = sorted((x.fp, x.rp, x.sz, x.dp, x.score) + tuple(int(np.int64(np.uint64(a))) for a in x.align)
                          for x in (hp[i] for i in range(max(nref, 0))))
            for search in (emu.emu_search, emu.emu_search_packed):
                op = C.POINTER(C.c_int64)()
                n = search(h, q.ctypes.data_as(C.c_void_p), len(q), tau, C.byref(op))
                if nref < 0:
                    assert n == -1
                    continue
                mine = sorted(tuple(int(op[10 * i + k]) for k in range(9)) for i in range(n))
                assert mine == want
                assert (int(op[9]) if n > 0 else int(op[0])) == ne.value
                emu.emu_free(op)
            if nref >= 0:
                po.lib()._libc.free(C.cast(hp, C.c_void_p))
    emu.emu_destroy(h)


def test_synthetic_generators_are_deterministic():
    a, _ = synth.make_genome(50_000, seed=9, n_seqs=3, repeat_rich=True)
    b, _ = synth.make_genome(50_000, seed=9, n_seqs=3, repeat_rich=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and sum(len(x) for x in a) == 50_000
    r1, o1 = synth.make_reads(a, 500, 60, 3, seed=4)
    r2, o2 = synth.make_reads(b, 500, 60, 3, seed=4)
    assert np.array_equal(r1, r2) and np.array_equal(o1, o2) and r1.shape == (500, 60)
    assert (r1 == 4).sum(axis=1).max() <= 1                  # no read is drawn from an N run
    s = synth.Stream(1)
    x = s.u64(10)
    s2 = synth.Stream(1)
    assert np.array_equal(np.concatenate([s2.u64(4), s2.u64(6)]), x)   # chunked == whole


def test_device_generator_on_cpu_tensors():
    """mapper_amd/synth_device.py with torch on the CPU: reads are windows of the sequences (reverse
    complemented where strand = 1) with at most max_mm substitutions and the occasional N; windows never
    span two sequences or contain a text N."""
    import torch
    from mapper_amd import synth_device
    g, lens, names = synth_device.make_genome(300_000, 3, n_seqs=3, device="cpu")
    assert g.numel() == 300_000 and int(lens.sum()) == 300_000 and len(names) == 3
    g2, _, _ = synth_device.make_genome(300_000, 3, n_seqs=3, device="cpu")
    assert torch.equal(g, g2)                                   # a function of the seed alone
    assert int((g == 4).sum()) >= 300_000 // 40
    reads, origin, strand = synth_device.make_reads(g, lens, 5000, 60, 2, seed=8, n_frac=0.05, return_origin=True)
    gn, rd, og, sd = g.numpy(), reads.numpy(), origin.numpy(), strand.numpy()
    comp = np.array([3, 2, 1, 0, 4], np.uint8)
    bounds = np.cumsum(lens)
    for r in range(0, 5000, 7):
        w = gn[og[r]:og[r] + 60]
        assert (w != 4).all()
        assert np.searchsorted(bounds, og[r], side="right") == np.searchsorted(bounds, og[r] + 59, side="right")
        if sd[r]:
            w = comp[w[::-1]]
        diff = (rd[r] != w) & (rd[r] != 4)
        assert diff.sum() <= 2 and (rd[r] == 4).sum() <= 1
