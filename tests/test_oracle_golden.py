"""Pins the oracle (oracle/fmoracle.c) against the reference: hit dumps produced by the unmodified
reference (tests/golden/*.bin, made by tools/make_golden.py) and the known-answer vectors of the
reference's own unit tests, transcribed as data (file:line cited at each block).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

import mapper_amd as ma
from oracle import pyoracle as po
from tests import common

A, C_, G, T, N = 0, 1, 2, 3, 4
SUFFIX, PREFIX = po.lib() and 1, 0


@pytest.fixture(scope="module")
def small(workdir):
    return po.Index(common.build_golden_index(workdir, "small"))


@pytest.fixture(scope="module")
def repeats(workdir):
    return po.Index(common.build_golden_index(workdir, "repeats"))


def test_index_files_byte_identical_to_reference(workdir):
    """The product's host builder writes the same four files as the reference's index_build."""
    base = common.build_golden_index(workdir, "small")
    want = json.load(open(os.path.join(common.GOLDEN, "small.sha256.json")))
    for ext, sha in want.items():
        assert hashlib.sha256(open(base + "." + ext, "rb").read()).hexdigest() == sha, ext


@pytest.mark.parametrize("tau", [0, 1, 2, 3, 4])
def test_oracle_equals_reference_dump_small(small, tau):
    reads = common.load_reads("small.reads.txt")
    codes, offs = common.encode_reads(reads)
    ref = po.HitDump.load(os.path.join(common.GOLDEN, "small.tau%d.bin" % tau))
    got = small.map_batch(codes, offs, tau, threads=4)
    assert np.array_equal(got.status, ref.status)
    assert np.array_equal(got.hits, ref.hits)          # same hits in the same (DFS) order
    assert np.array_equal(got.locs, ref.locs)


@pytest.mark.parametrize("tau", [0, 1, 2])
def test_oracle_equals_reference_dump_repeats(repeats, tau):
    reads = common.load_reads("repeats.reads.txt")
    codes, offs = common.encode_reads(reads)
    ref = po.HitDump.load(os.path.join(common.GOLDEN, "repeats.tau%d.bin" % tau))
    got = repeats.map_batch(codes, offs, tau, threads=1)
    assert np.array_equal(got.hits, ref.hits) and np.array_equal(got.locs, ref.locs)


def test_oracle_encode_is_sym_str_index(small):
    # src/index_sym.c:11-13,121-132: Aa Cc Gg Tt -> 0..3, everything else -> 4
    assert small.encode("ACGTNacgtnRYKM*- x").tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4] + [4] * 8
    reads = common.load_reads("small.reads.txt")
    codes, offs = common.encode_reads(reads)
    assert np.array_equal(np.concatenate([small.encode(r) for r in reads if r]), codes)


# ---------------------------------------------------------------- KATs of test/test_index_bwt.c

@pytest.fixture(scope="module")
def bwt32(workdir):
    """Text TTAGCAGTAGTCGTA$TACGACTACTGCTAA$ (test/test_index_bwt.c:14-17): one sequence plus its
    reverse complement, which is what index_build makes of a one-record FASTA."""
    base = os.path.join(workdir, "bwt32")
    ma.HostIndex.build_codes([common._TABLE[np.frombuffer(b"TTAGCAGTAGTCGTA", dtype=np.uint8)]], ["s"], base).close()
    return po.Index(base)


def _walk(ix, steps):
    """steps: list of (symbol, end); returns list of (fp, rp, sz) after each bwt_query."""
    fp, rp, sz = 0, 0, ix.txt_len
    out = []
    for sym, end in steps:
        fp, rp, sz = ix.extend_all(end, fp, rp, sz)[sym]
        out.append((fp, rp, sz))
    return out


def test_kat_suffix_array(bwt32):
    # test/test_index_bwt.c:19
    sa_ref = [31, 15, 30, 14, 29, 17, 20, 23, 2, 5, 8, 4, 18, 11, 27, 21, 24, 19, 3, 26, 12, 6, 9, 13, 28, 16, 22, 1,
              7, 10, 25, 0]
    assert bwt32.sar_get_range(0, 32) == sa_ref
    assert bwt32.sar_bits == 5


def test_kat_bwt_query_suffix_prefix(bwt32):
    ix = bwt32
    # test/test_index_bwt.c:41-85: G, GT, GTA, GTAG, GTAGC
    w = _walk(ix, [(G, 1), (T, 1), (A, 1), (G, 1), (C_, 1)])
    assert [(f, s) for f, _, s in w[:4]] == [(17, 6), (20, 3), (20, 2), (21, 1)] and w[4][2] == 0
    assert ix.sar_get_range(17, 6) == [19, 3, 26, 12, 6, 9]
    assert ix.sar_get_range(20, 3) == [12, 6, 9] and ix.sar_get_range(21, 1) == [6]
    # :91-132: T, TA, TAA; CTA by prefix
    w = _walk(ix, [(T, 1), (A, 1)])
    assert [(f, s) for f, _, s in w] == [(23, 9), (23, 6)]
    assert ix.sar_get_range(23, 9) == [13, 28, 16, 22, 1, 7, 10, 25, 0]
    fp, rp, sz = w[1]
    taa = ix.extend_all(1, fp, rp, sz)[A]
    assert (taa[0], taa[2]) == (24, 1) and ix.sar_get(24) == 28
    cta = ix.extend_all(0, fp, rp, sz)[C_]
    assert (cta[0], cta[2]) == (14, 2) and ix.sar_get_range(14, 2) == [27, 21]
    # :135-150: bwt_query_all PREFIX on TA
    qv = ix.extend_all(0, fp, rp, sz)
    assert [q[2] for q in qv] == [0, 2, 2, 1, 0]
    assert (qv[1][0], qv[2][0], qv[3][0]) == (14, 20, 31)
    # :154-182: TAC then all suffix extensions
    tac = ix.extend_all(1, fp, rp, sz)[C_]
    assert (tac[0], tac[2]) == (25, 2) and ix.sar_get_range(25, 2) == [16, 22]
    qv = ix.extend_all(1, *tac)
    assert [q[2] for q in qv] == [0, 0, 1, 1, 0] and (qv[2][0], qv[3][0]) == (25, 26)


def test_kat_reverse_pointers(bwt32):
    # test/test_index_bwt.c:578-635: fp, rp, sz after mixed-direction extensions
    assert _walk(bwt32, [(A, 1), (A, 1), (T, 0), (C_, 0)]) == [(2, 23, 9), (4, 31, 1), (24, 31, 1), (14, 31, 1)]
    assert _walk(bwt32, [(A, 0), (C_, 1), (T, 0), (T, 1), (C_, 0)]) == \
        [(2, 23, 9), (5, 20, 3), (25, 20, 2), (26, 9, 1), (15, 9, 1)]


def test_kat_query_all(bwt32):
    # test/test_index_bwt.c:684-727
    q0 = bwt32.extend_all(1, 0, 0, 32)
    assert [(q[0], q[2]) for q in q0[:4]] == [(2, 9), (11, 6), (17, 6), (23, 9)] and q0[4][2] == 0
    q1 = bwt32.extend_all(1, *q0[0])
    assert [(q[0], q[2]) for q in q1[:3]] == [(4, 1), (5, 3), (8, 3)] and q1[3][2] == 0 and q1[4][2] == 0
    q2 = bwt32.extend_all(1, *q1[2])
    assert [q[2] for q in q2] == [0, 1, 0, 2, 0] and (q2[1][0], q2[3][0]) == (8, 9)


def test_kat_prefix_all(bwt32):
    # test/test_index_bwt.c:187-238 (bwt_prefix / bwt_prefix_all): C, GC, XGC, TAGC
    c = bwt32.prefix_all(0, 32)[C_]
    assert c == (11, 6) and bwt32.sar_get_range(11, 6) == [4, 18, 11, 27, 21, 24]
    gc = bwt32.prefix_all(*c)[G]
    assert gc == (18, 2) and bwt32.sar_get_range(18, 2) == [3, 26]
    qv = bwt32.prefix_all(*gc)
    assert qv[0] == (8, 1) and qv[3] == (30, 1) and [q[1] for q in qv] == [1, 0, 0, 1, 0]
    tagc = bwt32.prefix_all(*qv[0])[T]
    assert tagc == (27, 1) and bwt32.sar_get(27) == 1


def test_kat_periodic_text_across_mark_intervals(workdir):
    # test/test_index_bwt.c:247-294: (ACGTTGCA$)x1000 (TGCAACGT$)x1000, 18000 symbols, 21 mark intervals
    base = os.path.join(workdir, "periodic")
    seq = common._TABLE[np.frombuffer(b"ACGTTGCA", dtype=np.uint8)]
    ma.HostIndex.build_codes([seq] * 1000, ["s%d" % i for i in range(1000)], base).close()
    ix = po.Index(base)
    assert ix.txt_len == 18000
    w = _walk(ix, [(A, 1), (C_, 1), (G, 1), (T, 1), (T, 1), (G, 1), (A, 1)])
    assert [(f, s) for f, _, s in w[:6]] == [(2000, 4000), (4000, 2000), (4000, 2000), (4000, 2000), (5000, 1000),
                                             (5000, 1000)]
    assert w[6][2] == 0


def test_kat_files_from_reference_primitives(bwt32, workdir):
    """kat_*.txt: the reference's sar_get / bwt_query_all answers on the two literal texts."""
    for name, ix in (("bwt32", bwt32), ("periodic", po.Index(os.path.join(workdir, "periodic")))):
        if name == "periodic" and not os.path.exists(os.path.join(workdir, "periodic.bwt")):
            pytest.skip("periodic index not built")
        lines = open(os.path.join(common.GOLDEN, "kat_%s.txt" % name)).read().split("\n")
        n = int(lines[0].split()[1])
        assert n == ix.txt_len
        assert [int(x) for x in lines[1].split()[1:]] == ix.sar_get_range(0, n)
        first = {}
        for ln in lines[2:]:
            if not ln:
                continue
            key, val = ln.split(" : ")
            k = [int(x) for x in key.split()[1:]]
            v = tuple(int(x) for x in val.split())
            if len(k) == 2:
                got = ix.extend_all(k[0], 0, 0, n)[k[1]]
                assert got == v[:3] and v[3] == 1
                first[(k[0], k[1])] = got
            else:
                got = ix.extend_all(k[2], *first[(k[0], k[1])])[k[3]]
                assert got == v[:3] and v[3] == 2


# ---------------------------------------------------------------- KATs of test/test_blocksearch.c

ONE, THREE, FIVE = "ATCGATATCAGCCACTACGAGACAA", "ATCGATATCAGCCACTACGAtACAA", "ATCGATATCAGCCACTACGAGACAc"
SEVEN, EIGHT, NONE = "ATCGATATaAGCCACTACGAGACAA", "ATCGATATtAGCCACTACGAGACAA", "NNNNATATCAGCCACTACGAGACAA"


def test_kat_blocksearch_trail_rec(repeats):
    ix = repeats
    # test/test_blocksearch.c:179-237 (blocks = 2 -> tau 1)
    def run(s):
        return ix.blocksearch(ix.encode(s), 1)
    h = run(FIVE)
    assert len(h) == 2
    for x in h:
        if x["score"] == 1:
            assert x["align"][0] == 1 << 24 and ix.pos_to_str(ix.sar_get(x["fp"])) == "one:1:+"
    h = run(THREE)
    assert len(h) == 2 and all(x["align"][0] == 1 << 20 for x in h if x["score"] == 1)
    h = run(SEVEN)
    assert len(h) == 3
    for x in h:
        if x["score"] == 1:
            assert x["align"][0] == 1 << 8 and ix.pos_to_str(ix.sar_get(x["fp"])) in ("one:1:+", "eight:1:+")
    assert len(run(ONE)) == 8
    assert len(run(NONE)) == 8          # a query N is a free wildcard (:231-236)


def test_kat_seqsearch_fw_bw(repeats):
    ix = repeats
    # test/test_blocksearch.c:294-434: seqsearch_fw(path, q, 0, 24, 1, 0, 0) and seqsearch_bw(path, q, 24, 0, 1, 0, 0)
    for s, n_hits, bit, loci in ((FIVE, 2, 24, ("one:1:+",)), (THREE, 2, 20, ("one:1:+",)),
                                 (SEVEN, 3, 8, ("one:1:+", "eight:1:+")), (ONE, 8, None, None), (NONE, 8, None, None)):
        q = ix.encode(s)
        for hits in (ix.seqsearch(1, {}, q, 0, 24, 1), ix.seqsearch(0, {}, q, 24, 0, 1)):
            assert len(hits) == n_hits
            if bit is not None:
                for x in hits:
                    if x["score"] == 1:
                        assert x["align"][0] == 1 << bit and ix.pos_to_str(ix.sar_get(x["fp"])) in loci


def test_kat_whole_index_walk(repeats):
    # test/test_index.c:24-47: 25 SUFFIX steps over "one" -> one:1:+ ; 25 PREFIX steps over "two" -> two:1:+
    ix = repeats
    one = ix.encode(ONE)
    w = _walk(ix, [(int(c), 1) for c in one])
    assert w[-1][2] == 1 and ix.pos_to_str(ix.sar_get(w[-1][0])) == "one:1:+"
    two = ix.encode("ATCGATATCAGgCACTACGAGACAA")
    w = _walk(ix, [(int(c), 0) for c in two[::-1]])
    assert w[-1][2] == 1 and ix.pos_to_str(ix.sar_get(w[-1][0])) == "two:1:+"


def test_kat_coordinates(repeats):
    # test/test_index_txt.c:598-612,651-655 rule: strand = pos >= n/2, pos' = n-2-pos, 1-based in sequence
    ix = repeats
    n = ix.txt_len
    assert ix.pos_to_str(0) == "one:1:+" and ix.pos_to_str(25) == "one:26:+" and ix.pos_to_str(26) == "two:1:+"
    assert ix.pos_to_str(n - 2) == "one:1:-" and ix.pos_to_str(n // 2) == "eight:25:-"
    with pytest.raises(ValueError):
        ix.pos_to_str(n)
