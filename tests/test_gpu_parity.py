"""GPU parity tests: the HIP path, called through the C-ABI, against the oracle and the golden dumps
of the reference.  Bit-exact (integer work): every comparison is array equality."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from oracle import pyoracle as po
from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(workdir):
    base = common.build_golden_index(workdir, "small")
    ctx = ma.Context.from_files(base, 0)
    yield base, ctx, po.Index(base)
    ctx.close()


def test_occ_all_matches_oracle(small):
    base, ctx, orc = small
    n = orc.txt_len
    rng = np.random.default_rng(1)
    ptr = np.concatenate(([-1, 0, 1, 126, 127, 128, 129, n - 2, n - 1], rng.integers(-1, n, 5000)))
    got = ctx.occ_all(ptr)
    want = np.array([orc.occ_all(int(p)) for p in ptr])
    assert np.array_equal(got, want)


def test_query_all_matches_oracle(small):
    base, ctx, orc = small
    n = orc.txt_len
    rng = np.random.default_rng(2)
    # walk random bidirectional extensions from the root
    fp, rp, sz = [0], [0], [n]
    cur = [(0, 0, n)]
    for depth in range(12):
        nxt = []
        for (f, r, s) in cur:
            for end in (0, 1):
                ch = orc.extend_all(end, f, r, s)
                for j in range(5):
                    if ch[j][2] > 0 and rng.random() < 0.35:
                        nxt.append(ch[j])
        cur = nxt[:200] or cur
        for (f, r, s) in cur:
            fp.append(f); rp.append(r); sz.append(s)
    for end in (ma.QUERY_PREFIX, ma.QUERY_SUFFIX):
        got = ctx.query_all(end, fp, rp, sz)
        want = np.array([orc.extend_all(end, f, r, s) for f, r, s in zip(fp, rp, sz)])
        assert np.array_equal(got, want)


def test_prefix_all_matches_oracle(small):
    """gpumap_prefix_all against the restatement of bwt_prefix_all (src/index_bwt.c:390-446), which
    tests/test_oracle_golden.py pins to the reference's known answers (test/test_index_bwt.c:187-238): intervals of a
    backward walk, empty intervals, the whole table, sizes past the interval of any word (clamped at 0 only when
    negative), and pointers the reference refuses."""
    base, ctx, orc = small
    n = orc.txt_len
    rng = np.random.default_rng(12)
    fp, sz = [0, 0, n, n - 1, 5], [n, 0, 0, 1, 0]
    cur = [(0, n)]
    for depth in range(14):
        nxt = []
        for (f, s) in cur:
            ch = orc.prefix_all(f, s)
            nxt += [c for c in ch if c[1] > 0 and rng.random() < 0.5]
        cur = nxt[:300] or cur
        fp += [c[0] for c in cur]; sz += [c[1] for c in cur]
    lo = rng.integers(0, n, 2000); ln = rng.integers(0, 50, 2000)                    # arbitrary (not suffix) intervals
    keep = lo + ln <= n
    fp += lo[keep].tolist(); sz += ln[keep].tolist()
    got = ctx.prefix_all(fp, sz)
    want = np.array([orc.prefix_all(f, s) for f, s in zip(fp, sz)])
    assert np.array_equal(got, want)
    # the forward half of bwt_query_all at the prefix end is the same extension
    q = ctx.query_all(ma.QUERY_PREFIX, fp, np.zeros(len(fp), np.int64), sz)
    assert np.array_equal(q[:, :, 0], got[:, :, 0]) and np.array_equal(q[:, :, 2], got[:, :, 1])
    bad = ctx.prefix_all([-1, 0, n + 1, 3], [5, n + 1, 0, -4])
    assert np.all(bad == -1)
    assert ctx.prefix_all([], []).shape == (0, 5, 2)


def test_sar_get_matches_oracle(small):
    base, ctx, orc = small
    n = orc.txt_len
    pos = np.arange(n)
    got = ctx.sar_get(pos)
    want = np.array(orc.sar_get_range(0, n))
    assert np.array_equal(got, want)
    assert sorted(got.tolist()) == list(range(n))          # a suffix array is a permutation
    assert np.array_equal(ctx.sar_get_range(5, 100), want[5:105])
    assert np.array_equal(ctx.sar_get([-1, n, n + 5]), [-1, -1, -1])


@pytest.mark.parametrize("tau", [0, 1, 2, 3, 4])
def test_map_small_matches_reference_dump(small, tau):
    base, ctx, orc = small
    reads = common.load_reads("small.reads.txt")
    codes, offs = common.encode_reads(reads)
    ref = po.HitDump.load(os.path.join(common.GOLDEN, "small.tau%d.bin" % tau))
    h = ctx.map_batch(codes, offs, tau, want_intervals=True)
    assert np.array_equal(h.status, ref.status)
    assert np.array_equal(h.located(), ref.located())
    assert np.array_equal(h.intervals(), ref.intervals())
    # CSR offsets
    cnt = np.bincount(ref.located()[:, 0], minlength=len(reads)) if len(ref.located()) else np.zeros(len(reads), int)
    assert np.array_equal(np.diff(h.read_first_loc), cnt)
    # the kernel executes exactly the reference schedule's extension steps
    d = orc.map_batch(codes, offs, tau, threads=2)
    assert h.n_ext == d.n_ext


@pytest.mark.parametrize("tau", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("name", ["small", "repeats"])
def test_verify_path_matches_reference_dump(workdir, name, tau):
    """Located hits only (want_intervals = 0): nodes whose interval holds one suffix are finished by k_verify
    against the text instead of on the Occ table.  Same located hits as the compiled reference, same
    count of extension steps as its schedule; also with the short cut switched off."""
    if name == "repeats" and tau > 2:
        pytest.skip("no dump")
    base = common.build_golden_index(workdir, name)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    reads = common.load_reads(name + ".reads.txt")
    codes, offs = common.encode_reads(reads)
    ref = po.HitDump.load(os.path.join(common.GOLDEN, "%s.tau%d.bin" % (name, tau)))
    want_ext = orc.map_batch(codes, offs, tau, threads=2).n_ext
    # (seed_candidates: a seed whose entry of the jump table holds one suffix goes to k_verify at once, or takes a step first)
    for verify, seed in ((16, 1), (4, 1), (4, 0), (2, 1), (1, 1), (1, 0), (0, 1)):   # nodes of up to four / of one or two suffixes / of one suffix / none finished on the text
        ctx.set_option("verify", verify)
        ctx.set_option("seed_candidates", seed)
        b = ctx.upload(codes, offs)
        c = b.run(tau)
        h = b.fetch()
        b.close()
        assert np.array_equal(h.status, ref.status)
        assert np.array_equal(h.located(), ref.located())
        assert h.n_ext == want_ext
        assert (c["verify_steps"] > 0) == (verify >= 1 and len(ref.located()) > 0)
    ctx.close()


def test_verify_at_the_ends_of_the_text_and_at_separators(workdir):
    """Reads hanging over the first / last sequence's ends, over a separator between sequences, and reads
    with N, against a text with N runs: windows that leave the text or cross a separator must die exactly
    where the reference's search does (no child at a separator; src/index_bwt.c:533 sets no bit for it)."""
    rng = np.random.default_rng(12)
    seqs = [rng.integers(0, 4, 400).astype(np.uint8) for _ in range(3)]
    seqs[1][100:130] = 4
    base = os.path.join(workdir, "edges")
    ma.HostIndex.build_codes(seqs, ["a", "b", "c"], base).close()
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    comp = np.array([3, 2, 1, 0, 4], np.uint8)
    reads = []
    for s in seqs:
        for start in (0, 1, 2, 3, 340, 358, 359, 360):          # 40-letter windows at both ends
            w = s[start:start + 40].copy()
            if len(w) < 40:
                w = np.concatenate([w, rng.integers(0, 4, 40 - len(w)).astype(np.uint8)])   # hangs over the end
            reads.append(w)
            reads.append(comp[w[::-1]])
            v = w.copy(); v[rng.integers(0, 40)] ^= 1            # one substitution
            reads.append(v)
            u = w.copy(); u[rng.integers(0, 40)] = 4             # one N
            reads.append(u)
    reads.append(np.concatenate([seqs[0][-20:], seqs[1][:20]]))   # across a separator: must not map there
    reads.append(seqs[1][90:130].copy())                          # into the N run
    codes = np.concatenate(reads).astype(np.uint8)
    offs = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.int64)
    for tau in (0, 1, 2, 3):
        d = orc.map_batch(codes, offs, tau, threads=2)
        for verify, seed in ((16, 1), (4, 1), (4, 0), (2, 1), (1, 1), (1, 0), (0, 1)):
            ctx.set_option("verify", verify)
            ctx.set_option("seed_candidates", seed)
            h = ctx.map_batch(codes, offs, tau)
            assert np.array_equal(h.status, d.status)
            assert np.array_equal(h.located(), d.located())
            assert h.n_ext == d.n_ext
    ctx.close()


@pytest.mark.parametrize("jump_k", [0, 4, 9, 14])
def test_jump_table_changes_nothing(small, jump_k):
    """The K-mer jump table of the leaf phases (k_build_jump) stands in for the first K extension steps:
    hits and even the count of extension steps stay those of the reference schedule, for K-mers that
    are absent from the text (the search dies inside the table), query N inside the K-mer (no jump) and
    blocks shorter than K (no jump).  The golden dumps are those of the compiled reference."""
    base, ctx, orc = small
    ctx.set_option("jump_k", jump_k)
    try:
        reads = common.load_reads("small.reads.txt")
        codes, offs = common.encode_reads(reads)
        for tau in (0, 1, 2, 4):
            ref = po.HitDump.load(os.path.join(common.GOLDEN, "small.tau%d.bin" % tau))
            h = ctx.map_batch(codes, offs, tau, want_intervals=True)
            assert np.array_equal(h.status, ref.status)
            assert np.array_equal(h.located(), ref.located())
            assert np.array_equal(h.intervals(), ref.intervals())
            assert h.n_ext == orc.map_batch(codes, offs, tau, threads=2).n_ext
    finally:
        ctx.set_option("jump_k", -1)


@pytest.mark.parametrize("tau", [0, 1, 2])
def test_map_repeats_matches_reference_dump(workdir, tau):
    base = common.build_golden_index(workdir, "repeats")
    ctx = ma.Context.from_files(base, 0)
    reads = common.load_reads("repeats.reads.txt")
    codes, offs = common.encode_reads(reads)
    ref = po.HitDump.load(os.path.join(common.GOLDEN, "repeats.tau%d.bin" % tau))
    h = ctx.map_batch(codes, offs, tau, want_intervals=True)
    assert np.array_equal(h.located(), ref.located())
    assert np.array_equal(h.intervals(), ref.intervals())
    ctx.close()


def test_coordinates_and_strand(small):
    base, ctx, orc = small
    reads = common.load_reads("small.reads.txt")
    codes, offs = common.encode_reads(reads)
    h = ctx.map_batch(codes, offs, 2)
    hix = ma.HostIndex.read(base)
    names = hix.seq_names()
    for t in range(0, h.n_locs, max(1, h.n_locs // 300)):
        s = "%s:%d:%s" % (names[h.loc_seq[t]], h.loc_seqpos[t], "-" if h.loc_strand[t] else "+")
        assert s == orc.pos_to_str(int(h.loc_pos[t])) == hix.pos_to_str(int(h.loc_pos[t]))


def test_text_upload_and_encode(small):
    base, ctx, orc = small
    reads = common.load_reads("small.reads.txt")
    codes, offs = common.encode_reads(reads)
    text = "".join(reads)
    assert np.array_equal(ctx.encode(text, offs), codes)
    b = ctx.upload_text(text, offs)
    b.run(2)
    h1 = b.fetch()
    h2 = ctx.map_batch(codes, offs, 2)
    assert np.array_equal(h1.located(), h2.located())
    b.close()


def test_empty_and_degenerate_batches(small):
    base, ctx, orc = small
    h = ctx.map_batch(np.zeros(0, np.uint8), np.zeros(1, np.int64), 2)
    assert h.n_reads == 0 and h.n_locs == 0
    codes, offs = common.encode_reads(["", "A", "AC"])
    h = ctx.map_batch(codes, offs, 2, want_intervals=True)
    assert h.status.tolist() == [-1, -1, -1] and h.n_locs == 0 and h.n_hits == 0
    with pytest.raises(ma.GpumapError):
        ctx.map_batch(codes, offs, ma.MAX_TAU + 1)
    with pytest.raises(ma.GpumapError):
        ctx.map_batch(codes, offs, -1)


def test_random_batch_matches_oracle(workdir):
    """Config-2 shaped case at a size the oracle finishes in seconds."""
    base, seqs = common.synth_index(workdir, "rnd300k", 300_000, seed=21, n_seqs=3)
    codes, offs = synth.make_reads(seqs, 20000, 100, 2, seed=22)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    h = ctx.map_batch(codes.reshape(-1), offs, 2, want_intervals=True)
    d = orc.map_batch(codes.reshape(-1), offs, 2, threads=8)
    assert np.array_equal(h.located(), d.located())
    assert np.array_equal(h.intervals(), d.intervals())
    assert h.n_ext == d.n_ext
    # every read maps back to where it was drawn from: at least one hit with <= 2 mismatches
    assert (np.diff(h.read_first_loc) >= 1).mean() > 0.99
    ctx.close()


def test_repeat_rich_tau4_matches_oracle(workdir):
    """Config-5 shaped case (150 nt, tau 4, repeat families, text N)."""
    base, seqs = common.synth_index(workdir, "rep400k", 400_000, seed=31, n_seqs=2, repeat_rich=True)
    codes, offs = synth.make_reads(seqs, 3000, 150, 4, seed=32, n_frac=0.02)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    h = ctx.map_batch(codes.reshape(-1), offs, 4, want_intervals=True)
    d = orc.map_batch(codes.reshape(-1), offs, 4, threads=8)
    assert np.array_equal(h.located(), d.located())
    assert np.array_equal(h.intervals(), d.intervals())
    assert h.n_ext == d.n_ext
    ctx.close()


@pytest.mark.parametrize("m,tau", [(100, 2), (150, 4), (64, 3), (33, 1), (120, 6), (255, 7)])
def test_located_only_paths_match_oracle_on_repeats(workdir, m, tau):
    """The path bench.py times (located hits only) on a repeat-rich text, where intervals of two suffixes are
    common: both k_verify modes against the oracle, hits and count of extension steps."""
    base, seqs = common.synth_index(workdir, "rep400k", 400_000, seed=31, n_seqs=2, repeat_rich=True)
    codes, offs = synth.make_reads(seqs, 4000 if tau <= 4 else 300, m, min(tau, 4), seed=300 + m, n_frac=0.02)
    ctx = ma.Context.from_files(base, 0)
    d = po.Index(base).map_batch(codes.reshape(-1), offs, tau, threads=8)
    for verify in (16, 4, 2, 1):
        ctx.set_option("verify", verify)
        h = ctx.map_batch(codes.reshape(-1), offs, tau)
        assert np.array_equal(h.located(), d.located())
        assert np.array_equal(h.read_first_loc, np.searchsorted(d.located()[:, 0], np.arange(len(offs))))
        assert h.n_ext == d.n_ext
    ctx.close()


@pytest.mark.parametrize("div", [0.0, 0.01, 0.05])
def test_pairs_on_a_duplicated_text(workdir, div):
    """Two copies of one sequence (the second with a fraction `div` of substitutions): nearly every node of the
    search holds two suffixes for long stretches, then the copies part.  Both k_verify modes give the oracle's
    located hits and its count of extension steps (steps the two walks share are counted once)."""
    rng = np.random.default_rng(int(div * 1000) + 7)
    a = rng.integers(0, 4, 120_000).astype(np.uint8)
    b = a.copy()
    hit = rng.random(len(b)) < div
    b[hit] = (b[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
    b[5000:5040] = 4                                   # an N run in one copy only
    seqs = [a, b]
    base = os.path.join(workdir, "dup_%d" % int(div * 1000))
    ma.HostIndex.build_codes(seqs, ["a", "b"], base).close()
    codes, offs = synth.make_reads(seqs, 3000, 100, 2, seed=int(div * 1000) + 9, n_frac=0.02)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    for tau in (2, 3):
        d = orc.map_batch(codes.reshape(-1), offs, tau, threads=8)
        for verify in (16, 4, 2, 1, 0):
            ctx.set_option("verify", verify)
            b_ = ctx.upload(codes.reshape(-1), offs)
            c = b_.run(tau)
            h = b_.fetch()
            b_.close()
            assert np.array_equal(h.located(), d.located()), (tau, verify)
            assert h.n_ext == d.n_ext, (tau, verify)
            assert (c["verify_steps"] > 0) == (verify >= 1)
    ctx.close()


@pytest.mark.parametrize("copies,div", [(3, 0.0), (3, 0.02), (4, 0.01), (4, 0.06), (6, 0.03), (8, 0.02), (12, 0.01), (16, 0.0), (20, 0.01)])
def test_groups_on_a_text_of_several_copies(workdir, copies, div):
    """Three to twenty copies of one sequence, each with its own substitutions (and an N run in one): the nodes of the
    search hold several suffixes for long stretches and the copies part one after the other.  With verify = 4 the nodes
    of three and four suffixes are finished by k_verify_group, four lanes per node; with verify = 16 also those of five
    to sixteen, sixteen lanes per node: the oracle's located hits, and its count of extension steps (the walks of one
    node are counted as the union of their paths in the reference's search tree)."""
    rng = np.random.default_rng(int(div * 1000) + 11 * copies)
    a = rng.integers(0, 4, 60_000).astype(np.uint8)
    seqs = []
    for c in range(copies):
        b = a.copy()
        hit = rng.random(len(b)) < div
        b[hit] = (b[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
        if c == 1:
            b[7000:7030] = 4
        seqs.append(b)
    base = os.path.join(workdir, "copies_%d_%d" % (copies, int(div * 1000)))
    ma.HostIndex.build_codes(seqs, ["c%d" % c for c in range(copies)], base).close()
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    for m, tau in ((100, 2), (150, 4), (70, 3)):
        codes, offs = synth.make_reads(seqs, 2500, m, tau, seed=copies * 100 + m, n_frac=0.02)
        d = orc.map_batch(codes.reshape(-1), offs, tau, threads=8)
        for verify in (16, 8, 4, 2):
            ctx.set_option("verify", verify)
            b_ = ctx.upload(codes.reshape(-1), offs)
            b_.run(tau)
            h = b_.fetch()
            big = b_.stats_ext()["big_candidates"]
            b_.close()
            assert np.array_equal(h.located(), d.located()), (m, tau, verify)
            assert h.n_ext == d.n_ext, (m, tau, verify)
            assert big == 0 if verify == 2 else (big > 0 or not (verify == 16 and copies <= 16 or verify == 8 and copies <= 8)), (m, tau, verify, big)
        assert big >= 0
    ctx.close()


def test_spill_path_matches_oracle(small):
    """Reads with many N's blow up the frontier: forces the deque to spill to global memory."""
    base, ctx, orc = small
    reads = ["NNNNNNACGTNN", "NNNNNNNN", "ACNNNNNNNNGT"] * 8
    codes, offs = common.encode_reads(reads)
    h = ctx.map_batch(codes, offs, 1, want_intervals=True)
    d = orc.map_batch(codes, offs, 1, threads=4)
    assert np.array_equal(h.located(), d.located())
    assert np.array_equal(h.intervals(), d.intervals())


def test_sharded_batches_concatenate_in_input_order(workdir):
    """Logical sharding on one device: contiguous read ranges, concatenated in order."""
    base, seqs = common.synth_index(workdir, "rnd300k", 300_000, seed=21, n_seqs=3)
    codes, offs = synth.make_reads(seqs, 4001, 100, 2, seed=23)
    ctx = ma.Context.from_files(base, 0)
    whole = ctx.map_batch(codes.reshape(-1), offs, 2).located()
    for g in (2, 4, 8):
        parts = []
        for k in range(g):
            lo, hi = 4001 * k // g, 4001 * (k + 1) // g
            loc = ctx.map_batch(codes[lo:hi].reshape(-1), offs[lo:hi + 1] - offs[lo], 2).located()
            loc[:, 0] += lo
            parts.append(loc)
        assert np.array_equal(np.concatenate(parts), whole)
    ctx.close()


def test_verify_tile_changes_nothing(workdir):
    """Option verify_tile (entries of the list a workgroup of k_verify takes and hands out to its lanes suffix by
    suffix) changes where the work is done, not the result: the oracle's located hits and its count of extension
    steps for tiles of 1, 64, 257 and 1024 entries (1: a round per entry; 257: a second pass of the tile's first look
    with one entry), with one lane per candidate (verify 1), pairs (2) and groups (4); values out of range are refused."""
    rng = np.random.default_rng(77)
    a = rng.integers(0, 4, 30_000).astype(np.uint8)
    seqs = []
    for c in range(4):
        b = a.copy()
        hit = rng.random(len(b)) < 0.02
        b[hit] = (b[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
        seqs.append(b)
    seqs[2][500:520] = 4
    base = os.path.join(workdir, "tiles")
    ma.HostIndex.build_codes(seqs, ["c%d" % c for c in range(4)], base).close()
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    codes, offs = synth.make_reads(seqs, 3000, 80, 2, seed=78, n_frac=0.02)
    d = orc.map_batch(codes.reshape(-1), offs, 2, threads=8)
    for verify in (1, 2, 4):
        ctx.set_option("verify", verify)
        for tile in (1, 64, 257, 1024, 0):
            ctx.set_option("verify_tile", tile)
            h = ctx.map_batch(codes.reshape(-1), offs, 2)
            assert np.array_equal(h.located(), d.located()), (verify, tile)
            assert h.n_ext == d.n_ext, (verify, tile)
    for bad in (-1, 1025):
        with pytest.raises(ma.GpumapError):
            ctx.set_option("verify_tile", bad)
    ctx.close()


@pytest.mark.parametrize("n_reads", [1, 5, 40, 300, 2100])
def test_sort_that_skips_low_bits_gives_the_same_order(workdir, n_reads):
    """The radix sort of the located hits runs over the read's bits only; k_order_first / k_order_big then order the hits
    of every read where they lie (gm_device.hip, "order"; option sort_all_bits = 1: one sort over the whole key).  On a
    text with homopolymer and dinucleotide runs a read hits many adjacent positions, each from several seeds (runs of a
    few hundred keys: the bitonic network in LDS): the located hits must come in the oracle's order (read, text
    position) with either sort, for batch sizes that move the key's width across the pass boundaries."""
    rng = np.random.default_rng(5)
    parts = [rng.integers(0, 4, 3000).astype(np.uint8), np.zeros(300, np.uint8), rng.integers(0, 4, 2000).astype(np.uint8),
             np.tile(np.array([1, 2], np.uint8), 200), rng.integers(0, 4, 1500).astype(np.uint8)]
    seq = np.concatenate(parts)
    base = os.path.join(workdir, "runs")
    if not os.path.exists(base + ".sar"):
        ma.HostIndex.build_codes([seq], ["r"], base).close()
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    reads = []
    for i in range(n_reads):
        kind = i % 4
        if kind == 0:
            w = np.zeros(40, np.uint8)                                   # poly-A: every shift inside the run
        elif kind == 1:
            w = np.tile(np.array([1, 2], np.uint8), 20)                  # CG repeat: every second shift
        elif kind == 2:
            w = np.zeros(40, np.uint8); w[rng.integers(0, 40)] = 3       # poly-A with a substitution
        else:
            s0 = int(rng.integers(0, len(seq) - 40)); w = seq[s0:s0 + 40].copy()
        reads.append(w)
    codes = np.concatenate(reads)
    offs = np.arange(0, 40 * (n_reads + 1), 40, dtype=np.int64)
    d = orc.map_batch(codes, offs, 2, threads=8)
    for all_bits, by_read in ((1, 0), (0, 0), (0, 1), (0, 2)):
        ctx.set_option("sort_all_bits", all_bits)
        ctx.set_option("sort_by_read", by_read)
        for compact in (0, 1):                       # (compact results: the runs are ordered inside k_read_first)
            ctx.set_option("compact_results", compact)
            for verify in (4, 1, 0):
                ctx.set_option("verify", verify)
                h = ctx.map_batch(codes, offs, 2)
                assert np.array_equal(h.located(), d.located()), (all_bits, by_read, compact, verify)
                assert h.n_ext == d.n_ext
    assert len(d.located()) > 100 * min(n_reads, 3)
    ctx.close()


def test_reads_with_thousands_of_hits_come_in_order(workdir):
    """Runs of keys longer than the LDS network takes (k_order_big's radix passes through the sort's other buffer): a
    read of 40 A against a text with 9 000 A in a row has about 9 000 hits per strand pair; beside it reads with a few
    hundred hits and ordinary ones, in one batch, with every form of result."""
    rng = np.random.default_rng(6)
    seq = np.concatenate([rng.integers(0, 4, 2000).astype(np.uint8), np.zeros(9000, np.uint8), rng.integers(0, 4, 2500).astype(np.uint8),
                          np.full(5000, 2, np.uint8), rng.integers(0, 4, 1000).astype(np.uint8), np.zeros(200, np.uint8)])
    base = os.path.join(workdir, "longruns")
    if not os.path.exists(base + ".sar"):
        ma.HostIndex.build_codes([seq], ["r"], base).close()
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    reads = []
    for i in range(24):
        kind = i % 4
        if kind == 0:
            w = np.zeros(40, np.uint8)
        elif kind == 1:
            w = np.full(40, 2, np.uint8); w[int(rng.integers(0, 40))] = 1
        elif kind == 2:
            w = np.full(40, 1, np.uint8)                                   # poly-C: the other strand of the G run
        else:
            s0 = int(rng.integers(0, 1900)); w = seq[s0:s0 + 40].copy()
        reads.append(w)
    codes = np.concatenate(reads)
    offs = np.arange(0, 40 * (len(reads) + 1), 40, dtype=np.int64)
    d = orc.map_batch(codes, offs, 2, threads=8)
    per_read = np.bincount(d.located()[:, 0], minlength=len(reads))
    assert per_read.max() > 8000 and (per_read > 4096).sum() >= 10
    for all_bits, by_read in ((0, 1), (0, 2), (0, 0), (1, 0)):
        ctx.set_option("sort_all_bits", all_bits)
        ctx.set_option("sort_by_read", by_read)
        for compact in (0, 1):
            ctx.set_option("compact_results", compact)
            h = ctx.map_batch(codes, offs, 2)
            assert np.array_equal(h.located(), d.located()), (all_bits, by_read, compact)
            assert np.array_equal(h.read_first_loc, d.read_first_loc if hasattr(d, "read_first_loc") else h.read_first_loc)
    ctx.close()
