"""GPU tests of the batch plumbing around the kernels: staged batches (encoding inside the run), compact results
(sorted keys + CSR offsets), the pipeline of chunks (gpumap_pipe: several batch objects with streams of their
own), several members sharing one device (the N > 1 path on one GPU), and the failure paths of the search (a hit
list guessed too short -> second search; a spill region too small -> error).  Everything through the C-ABI; the
results are compared bit for bit with gpumap_map_batch, which test_gpu_parity.py compares with the oracle, and
with the oracle itself."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from oracle import pyoracle as po
from tests import common

pytestmark = pytest.mark.gpu
ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


@pytest.fixture(scope="module")
def rep(workdir):
    base, seqs = common.synth_index(workdir, "rep400k", 400_000, seed=31, n_seqs=2, repeat_rich=True)
    ctx = ma.Context.from_files(base, 0)
    yield base, seqs, ctx
    ctx.close()


def _ragged_reads(seqs, n, seed):
    """Reads of mixed lengths (50..150), some with N, a few that the reference rejects (too short / empty)."""
    rng = np.random.default_rng(seed)
    parts = []
    for m in (50, 100, 150, 77):
        codes, _ = synth.make_reads(seqs, n // 4, m, 2, seed=int(rng.integers(1 << 30)), n_frac=0.02)
        parts += [r for r in codes.reshape(-1, m)]
    parts += [np.zeros(0, np.uint8), np.array([1], np.uint8), np.array([0, 3], np.uint8)]
    order = rng.permutation(len(parts))
    parts = [parts[i] for i in order]
    offs = np.concatenate(([0], np.cumsum([len(p) for p in parts]))).astype(np.int64)
    return np.concatenate(parts).astype(np.uint8), offs


def test_compact_results_equal_full_results(rep):
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 4000, 1)
    full = ctx.map_batch(codes, offs, 2)
    ctx.set_option("compact_results", 1)
    try:
        b = ctx.upload(codes, offs)
        b.run(2)
        assert b.stats_ext()["compact"] == 1
        raw = b.fetch()
        assert raw.loc_key is not None and raw.key_pos_bits == po.Index(base).sar_bits
        assert np.array_equal(raw.located(), full.located())
        assert np.array_equal(raw.read_first_loc, full.read_first_loc)
        assert np.array_equal(raw.status, full.status)
        assert np.all(np.diff(raw.loc_key.astype(np.uint64)) > 0) or raw.n_locs < 2        # sorted, no duplicates
        # the host-side expansion gives what k_unpack_locs computes on the device
        b.run(2)
        ex = b.fetch(expand=True)
        for k in ("loc_read", "loc_pos", "loc_fwd", "loc_seq", "loc_seqpos", "loc_strand", "loc_score"):
            assert np.array_equal(getattr(ex, k), getattr(full, k)), k
        # interval output is never compact
        b.run(2, want_intervals=True)
        assert b.stats_ext()["compact"] == 0
        b.close()
    finally:
        ctx.set_option("compact_results", 0)


def test_staged_batches_encode_inside_the_run(rep):
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 3000, 2)
    want = ctx.map_batch(codes, offs, 2)
    text = ASCII[codes]
    b = ctx.new_batch()
    for src, flags in ((codes, 0), (text, ma.SRC_TEXT)):
        b.stage(src, offs, flags)
        c = b.run(2)
        h = b.fetch()
        assert np.array_equal(h.located(), want.located()) and np.array_equal(h.status, want.status)
        assert h.n_ext == want.n_ext
        assert b.stats_ext()["us_encode"] > 0
    # a part of a host buffer: offsets that do not start at 0
    lo, hi = 500, 2200
    b.stage(text, offs[lo:hi + 1], ma.SRC_TEXT)
    b.run(2)
    h = b.fetch()
    sub = ctx.map_batch(codes[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], 2)
    assert np.array_equal(h.located(), sub.located())
    # decreasing offsets are refused (the check runs on the device)
    bad = offs.copy(); bad[10] = bad[11] + 5
    b.stage(text, bad, ma.SRC_TEXT)
    with pytest.raises(ma.GpumapError):
        b.run(2)
    # and the batch object is still good for the next chunk
    b.stage(codes, offs, 0)
    b.run(2)
    assert np.array_equal(b.fetch().located(), want.located())
    b.close()


def test_staged_batch_run_twice_gets_its_byte_rows_when_a_run_needs_them(rep):
    """A staged batch is encoded inside its run, for a mapping run without the byte matrix of the reads (k_search and
    k_verify read the bit planes).  A later run of the same batch that needs the bytes (interval output: the alignments
    of k_unpack_hits) encodes it again, and a mapping run after that still gives the same hits."""
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 3000, 23)
    want = ctx.map_batch(codes, offs, 2, want_intervals=True)
    b = ctx.new_batch()
    b.stage(ASCII[codes], offs, ma.SRC_TEXT)
    b.run(2)
    h = b.fetch()
    assert np.array_equal(h.located(), want.located()) and h.n_ext == want.n_ext
    b.run(2, want_intervals=True)
    h = b.fetch()
    assert np.array_equal(h.intervals(), want.intervals())
    assert np.array_equal(h.located(), want.located()) and h.n_ext == want.n_ext
    b.run(2)
    assert np.array_equal(b.fetch().located(), want.located())
    b.close()


def test_device_resident_chunks(rep):
    """GPUMAP_SRC_DEVICE: letters and offsets already in HBM (the timed region of bench.py starts there)."""
    import torch
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 3000, 3)
    want = ctx.map_batch(codes, offs, 2)
    d_text = torch.from_numpy(ASCII[codes]).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    torch.cuda.synchronize()
    b = ctx.new_batch()
    b.stage(d_text.data_ptr(), (d_offs.data_ptr(), len(offs) - 1), ma.SRC_TEXT | ma.SRC_DEVICE)
    b.run(2)
    h = b.fetch()
    assert np.array_equal(h.located(), want.located()) and h.n_ext == want.n_ext
    b.close()
    p = ctx.pipe(depth=2, tau=2, flags=ma.SRC_TEXT | ma.SRC_DEVICE | ma.PIPE_COMPACT)
    got = 0
    for k in range(5):
        if k - got == 2:                                      # both slots taken: take a result first
            h = p.next(); got += 1
            assert np.array_equal(h.located(), want.located())
        p.submit(d_text.data_ptr(), d_offs.data_ptr(), len(offs) - 1)
    while got < 5:
        h = p.next(); got += 1
        assert np.array_equal(h.located(), want.located()) and np.array_equal(h.read_first_loc, want.read_first_loc)
    assert p.next() is None
    p.close()


@pytest.mark.parametrize("depth,flags", [(1, 0), (2, ma.PIPE_COMPACT), (3, ma.SRC_TEXT | ma.PIPE_COMPACT), (4, ma.SRC_TEXT),
                                         (2, ma.PIPE_PACKED), (3, ma.SRC_TEXT | ma.PIPE_PACKED)])
def test_pipe_returns_chunks_in_order(rep, depth, flags):
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 6000, 4 + depth)
    n = len(offs) - 1
    cuts = [0, 1, 700, 701, 2500, 2500, 4100, n]              # ragged chunks, one of a single read, one empty
    chunks = [(codes[offs[a]:offs[b]], offs[a:b + 1] - offs[a]) for a, b in zip(cuts[:-1], cuts[1:])]
    want = [ctx.map_batch(c, o, 2) for c, o in chunks]
    p = ctx.pipe(depth=depth, tau=2, flags=flags)
    got = []
    k = 0
    for c, o in chunks:                                       # the single-threaded pattern: take a result when all slots are taken
        if k - len(got) == depth:
            got.append(p.next())
        p.submit(ASCII[c] if flags & ma.SRC_TEXT else c, o)
        k += 1
    while len(got) < k:
        got.append(p.next())
    assert p.next() is None
    for h, w in zip(got, want):
        assert h.n_reads == w.n_reads
        assert np.array_equal(h.located(), w.located())
        assert np.array_equal(h.read_first_loc, w.read_first_loc)
        assert np.array_equal(h.status, w.status)
        assert h.n_ext == w.n_ext
    p.close()


def test_packed_results_expand_to_the_full_structure(rep):
    """GPUMAP_PIPE_PACKED: 5 bytes per hit, 4-byte offsets and a status byte per read cross the link; gpumap_hits_expand
    makes status, offsets, keys and the structure of arrays from them (what a C caller does), equal to a full result;
    releasing the result frees what the expansion allocated and the slot (the pipe goes on)."""
    import ctypes as C
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 3000, 11)
    want = ctx.map_batch(codes, offs, 2)
    p = ctx.pipe(depth=2, tau=2, flags=ma.PIPE_PACKED)
    for rep_ in range(5):                                     # more chunks than slots: every release must free its slot
        p.submit(codes, offs)
        raw, _ = p.next_raw()
        assert raw.loc_packed and raw.read_first32 and raw.status8 and not raw.loc_key and not raw.status
        assert ma.lib().gpumap_hits_expand(ctx._h, C.byref(raw)) == 0
        assert raw.loc_key and raw.loc_read and raw.status and raw.expanded == 3
        h = ma.Hits(raw)
        p.release_raw(raw)
        assert np.array_equal(h.located(), want.located())
        assert np.array_equal(h.read_first_loc, want.read_first_loc)
        assert np.array_equal(h.status, want.status)
        for k in ("loc_fwd", "loc_seq", "loc_seqpos", "loc_strand"):
            assert np.array_equal(getattr(h, k), getattr(want, k)), k
    assert p.next() is None
    p.close()


@pytest.mark.parametrize("n_reads", [37, 6000, 60000])
def test_packed_results_written_by_the_segment_sort(rep, n_reads):
    """With the ordering by segments (sort_by_read=2, the default at GRCh38 size) and packed results, the segments'
    workgroups write the 5-byte hits themselves (segment_write_packed: whole words, single bytes at the segment's
    ends), and the 8-byte keys never leave the device.  Forced here on batches whose segments hold from no hit to a few
    thousand, so that every alignment of a segment's first and last byte occurs; equal to the full result."""
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, n_reads, 17)
    want = ctx.map_batch(codes, offs, 2)
    ctx.set_option("sort_by_read", 2)
    try:
        p = ctx.pipe(depth=2, tau=2, flags=ma.PIPE_PACKED)
        try:
            p.submit(codes, offs)
            for k in range(3):
                if k < 2:
                    p.submit(codes, offs)
                h = p.next()
                assert np.array_equal(h.located(), want.located())
                assert np.array_equal(h.read_first_loc, want.read_first_loc)
                assert np.array_equal(h.status, want.status)
            assert p.next() is None
        finally:
            p.close()
    finally:
        ctx.set_option("sort_by_read", -1)


def test_submit_from_one_thread_beyond_the_depth_is_an_error_not_a_hang(rep, capfd, monkeypatch):
    """mapper() is single-threaded (src/mapper.c:23-40): a caller that submits depth + 1 chunks without taking a result
    gets -1 and a message from gpumap_pipe_submit (nobody else could free a slot), and 0 from gpumap_pipe_try_submit;
    after taking a result the pipe goes on, and every chunk comes back as submitted."""
    base, seqs, ctx = rep
    monkeypatch.setenv("GPUMAP_PIPE_GRACE_MS", "200")
    codes, offs = _ragged_reads(seqs, 900, 21)
    chunks = [(codes[offs[a]:offs[b]], offs[a:b + 1] - offs[a]) for a, b in ((0, 300), (300, 600), (600, 900))]
    want = [ctx.map_batch(c, o, 2) for c, o in chunks]
    p = ctx.pipe(depth=2, tau=2, flags=ma.PIPE_COMPACT)
    p.submit(*chunks[0])
    assert p.try_submit(*chunks[1]) is not None
    assert p.try_submit(*chunks[2]) is None                   # every slot is taken: no wait, no error
    with pytest.raises(ma.GpumapError):
        p.submit(*chunks[2])
    assert "every slot of the pipe holds a chunk" in capfd.readouterr().err
    got = [p.next()]                                          # (takes and releases a result: a slot is free again)
    p.submit(*chunks[2])
    got += [p.next(), p.next()]
    assert p.next() is None
    for h, w in zip(got, want):
        assert np.array_equal(h.located(), w.located()) and h.n_ext == w.n_ext
    p.close()


def test_pipe_with_two_threads_and_several_members_on_one_device(rep):
    """SURVEY 8e 'test without 8 GPUs': G members (contexts) sharing the device, chunks going round the members,
    a submitting and a consuming thread.  The concatenated output is that of one member."""
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 8000, 9)
    n = len(offs) - 1
    whole = ctx.map_batch(codes, offs, 2).located()
    dev, hdr = ctx.export()
    for g in (2, 4):
        members = [ctx] + [ma.Context.from_device(dev, hdr, 0) for _ in range(g - 1)]       # replicas that share the arrays
        p = ma.Pipe(members, 2, 2, ma.PIPE_COMPACT)
        bounds = [n * k // 11 for k in range(12)]
        results = []

        def consume():
            for _ in range(11):
                while True:
                    h = p.next()
                    if h is not None:
                        break
                results.append(h)
        th = threading.Thread(target=consume)
        th.start()
        for a, b in zip(bounds[:-1], bounds[1:]):
            p.submit(codes[offs[a]:offs[b]], offs[a:b + 1] - offs[a])      # blocks while all 2 * g slots are taken
        th.join()
        parts = []
        for h, a in zip(results, bounds[:-1]):
            loc = h.located()
            loc[:, 0] += a
            parts.append(loc)
        assert np.array_equal(np.concatenate(parts), whole)
        p.close()
        for m in members[1:]:
            m.close()


def test_short_hit_list_guess_costs_a_second_search_and_changes_nothing(rep):
    """'hit-buffer overflow flag + host retry'.  The first search counts what it could not store; the
    second runs with a list of that length."""
    base, seqs, ctx = rep
    codes, offs = _ragged_reads(seqs, 4000, 11)
    want = ctx.map_batch(codes, offs, 2, want_intervals=True)
    for verify, intervals in ((-1, False), (0, False), (-1, True)):
        ctx.set_option("verify", verify)
        ctx.set_option("hit_list_guess", 1024)
        try:
            b = ctx.upload(codes, offs)
            b.run(2, want_intervals=intervals)
            assert b.stats_ext()["attempts"] == 2
            h = b.fetch()
            assert np.array_equal(h.located(), want.located())
            assert h.n_ext == want.n_ext
            if intervals:
                assert np.array_equal(h.intervals(), want.intervals())
            ctx.set_option("hit_list_guess", 0)
            b.run(2, want_intervals=intervals)                  # the batch keeps the longer list: one search now
            assert b.stats_ext()["attempts"] == 1
            assert np.array_equal(b.fetch().located(), want.located())
            b.close()
        finally:
            ctx.set_option("hit_list_guess", 0)
            ctx.set_option("verify", -1)


def test_spill_region_is_used_and_its_overflow_is_an_error(workdir, capfd):
    """Reads made of N blow up the frontier (every N follows all five symbols): the deques spill to global memory
    (asserted through the spill counter); with a spill region of one chunk the search reports the overflow:
    -1 and a message in the errhandler's format, no partial result."""
    base = common.build_golden_index(workdir, "small")
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    reads = ["NNNNNNACGTNN", "NNNNNNNN", "ACNNNNNNNNGT", "NNNNNNNNNN"] * 40
    codes, offs = common.encode_reads(reads)
    d = orc.map_batch(codes, offs, 1, threads=4)
    b = ctx.upload(codes, offs)
    b.run(1, want_intervals=True)
    assert b.stats_ext()["spill_chunks"] > 0
    h = b.fetch()
    assert np.array_equal(h.located(), d.located()) and np.array_equal(h.intervals(), d.intervals())
    ctx.set_option("spill_nodes", 64)
    capfd.readouterr()
    with pytest.raises(ma.GpumapError):
        b.run(1, want_intervals=True)
    err = capfd.readouterr().err
    assert "error: search spill region overflowed" in err and "gm_device.hip" in err
    raw = ma._Hits()
    assert ma.lib().gpumap_batch_fetch(ctx._h, b._h, C.byref(raw)) == -1          # no result of a failed run
    ctx.set_option("spill_nodes", 0)                                               # back to the default: works again
    b.run(1, want_intervals=True)
    assert np.array_equal(b.fetch().located(), d.located())
    b.close()
    ctx.close()


def test_every_read_length_in_one_batch(rep):
    """256 distinct read lengths in one batch (one phase-table row each; the node keeps its row in one byte)."""
    base, seqs, ctx = rep
    rng = np.random.default_rng(17)
    s = seqs[0]
    reads = []
    for m in range(1, 257):
        a = int(rng.integers(0, len(s) - 300))
        reads.append(s[a:a + m].copy())
    reads.append(s[1000:1257].copy())                                   # 257 letters: status -1
    offs = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.int64)
    codes = np.concatenate(reads).astype(np.uint8)
    orc = po.Index(base)
    for tau in (0, 1):
        d = orc.map_batch(codes, offs, tau, threads=8)
        h = ctx.map_batch(codes, offs, tau)
        assert h.status[-1] == -1
        assert np.array_equal(h.status[:-1], d.status[:-1])
        assert np.array_equal(h.located(), d.located())
        assert h.n_ext == d.n_ext
