"""Pins the oracle's restatement of blocksc_trail / scsearch_fw (src/blocksearch.c:7-103, :314-378; the search
ann_build_mt runs for every k-mer of the genome, src/index_ann.c:556-567) against the reference: hit dumps of the
unmodified reference (tests/golden/*.sc.*.bin, made by tools/make_golden_sc.py with oracle/_ref/ref_driver) and
the known answers of test/test_blocksearch.c:65-116 and :497-594, transcribed as data.  CPU only."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import common

FIXTURES = [("repeats", "repeats.sc.reads.txt", "repeats.sc.tau%d.bin", (0, 1, 2)),
            ("small", "small.sc.k24.txt", "small.sc.k24.tau%d.bin", (1, 2, 3)),
            ("small", "small.sc.k31.txt", "small.sc.k31.tau%d.bin", (1, 2, 3)),
            ("dense", "dense.sc.k12.txt", "dense.sc.k12.tau%d.bin", (1, 2, 3)),
            ("dense", "dense.sc.k15.txt", "dense.sc.k15.tau%d.bin", (1, 2, 3))]
CASES = [(ix, q, d % t, t) for ix, q, d, taus in FIXTURES for t in taus]


@pytest.mark.parametrize("index,queries,dump,tau", CASES)
def test_oracle_blocksc_equals_reference_dump(workdir, index, queries, dump, tau):
    orc = po.Index(common.build_golden_index(workdir, index))
    codes, offs = common.encode_reads(common.load_reads(queries))
    ref = po.HitDump.load(os.path.join(common.GOLDEN, dump))
    got = orc.blocksc_batch(codes, offs, tau, threads=3)
    assert np.array_equal(got.status, ref.status)
    assert np.array_equal(got.hits, ref.hits)          # same hits in the same order (DFS order, after the reference's swap-removal)
    assert np.array_equal(got.locs, ref.locs)


@pytest.mark.parametrize("k,tau", [(24, 1), (24, 2), (24, 3), (31, 1), (31, 2), (31, 3)])
def test_trail_reuse_of_the_caller_changes_no_hit_set(k, tau):
    """ann_build_mt keeps a tree per strand class and tells blocksc_trail how many letters the k-mer shares with the
    last one of its class (`trail`): the left block's hits are then reused instead of searched again.  The batched
    entry searches every k-mer afresh (trail = 0).  The two dumps of the reference hold the same hits per k-mer, except
    where the reference's own reuse is inexact: the reused hits were computed for another mismatch budget (a k-mer
    with N lowers tau, src/blocksearch.c:57-70).  Those k-mers are named here, and they all touch an N."""
    queries = common.load_reads("small.sc.k%d.txt" % k)
    fresh = po.HitDump.load(os.path.join(common.GOLDEN, "small.sc.k%d.tau%d.bin" % (k, tau)))
    trail = po.HitDump.load(os.path.join(common.GOLDEN, "small.sctrail.k%d.tau%d.bin" % (k, tau)))
    assert np.array_equal(fresh.status, trail.status)

    def per_query(d):
        out = {}
        for h in d.intervals():
            out.setdefault(int(h[0]), []).append(tuple(int(x) for x in h[1:]))
        return out
    a, b = per_query(fresh), per_query(trail)
    differ = sorted(r for r in set(a) | set(b) if a.get(r) != b.get(r))
    assert len(differ) <= len(queries) // 100
    with_n = {i for i, q in enumerate(queries) if "N" in q.upper()}
    for r in differ:                                       # the k-mer itself or one of the ten before it holds an N
        assert any(j in with_n for j in range(max(0, r - 10), r + 1)), r


@pytest.fixture(scope="module")
def repeats(workdir):
    return po.Index(common.build_golden_index(workdir, "repeats"))


ONE = "ATCGATATCAGCCACTACGAGACAA"
THREE = "ATCGATATCAGCCACTACGAtACAA"
FIVE = "ATCGATATCAGCCACTACGAGACAc"
SEVEN = "ATCGATATaAGCCACTACGAGACAA"
EIGHT = "ATCGATATtAGCCACTACGAGACAA"
NONE = "NNNNATATCAGCCACTACGAGACAA"


def _locus(orc, hit):
    return orc.pos_to_str(orc.sar_get(hit["fp"]))


def test_kat_blocksc_trail(repeats):
    """test/test_blocksearch.c:64-116 (tau 1, the reference's trail arguments 0 / 15 give the same sets)."""
    # :65-74 FIVE: two hits, the one with a mismatch is "one" with bit 24 set
    hits = repeats.blocksc(repeats.encode(FIVE), 1)
    assert len(hits) == 2
    for h in hits:
        if h["score"] == 1:
            assert h["align"][0] == 1 << 24 and _locus(repeats, h) == "one:1:+"
    # :84-93 THREE
    hits = repeats.blocksc(repeats.encode(THREE), 1)
    assert len(hits) == 2
    for h in hits:
        if h["score"] == 1:
            assert h["align"][0] == 1 << 20 and _locus(repeats, h) == "one:1:+"
    # :103-104 SEVEN: the mismatch is in the first half, only itself is found
    assert len(repeats.blocksc(repeats.encode(SEVEN), 1)) == 1
    # :114-115 NONE: four N lower tau below zero
    assert repeats.blocksc(repeats.encode(NONE), 1) == []
    # :59-60 invalid arguments
    assert repeats.blocksc(np.zeros(0, np.uint8), 1) is None
    assert repeats.blocksc(repeats.encode(FIVE), -1) is None


def test_kat_scsearch_fw(repeats):
    """test/test_blocksearch.c:495-594: scsearch_fw from the root interval over the whole query, tau 1, boundary 1."""
    root = {}
    # :497-506 FIVE
    hits = repeats.scsearch(root, repeats.encode(FIVE), 0, 24, 1)
    assert len(hits) == 2
    for h in hits:
        if h["score"] == 1:
            assert h["align"][0] == 1 << 24 and _locus(repeats, h) == "one:1:+"
    # :517-526 THREE
    hits = repeats.scsearch(root, repeats.encode(THREE), 0, 24, 1)
    assert len(hits) == 2
    for h in hits:
        if h["score"] == 1:
            assert h["align"][0] == 1 << 20 and _locus(repeats, h) == "one:1:+"
    # :537-538 SEVEN
    assert len(repeats.scsearch(root, repeats.encode(SEVEN), 0, 24, 1)) == 1
    # :549-558 EIGHT
    hits = repeats.scsearch(root, repeats.encode(EIGHT), 0, 24, 1)
    assert len(hits) == 3
    for h in hits:
        if h["score"] == 1:
            assert h["align"][0] == 1 << 8 and _locus(repeats, h) in ("one:1:+", "seven:1:+")
    # :569-578 ONE
    hits = repeats.scsearch(root, repeats.encode(ONE), 0, 24, 1)
    assert len(hits) == 3
    for h in hits:
        if h["score"] == 1:
            assert h["align"][0] in (1 << 8, 1 << 10) and _locus(repeats, h) in ("six:1:+", "seven:1:+")
    # :589-590 NONE through seqsearch_fw: N is a wildcard, all eight sequences
    assert len(repeats.seqsearch(1, root, repeats.encode(NONE), 0, 24, 1)) == 8
