"""GPU parity of gpumap_ann_build (ann_build, src/index_ann.c:56-257; SURVEY 8f rank 4: the reference's consumer of
blocksc_trail) through the C-ABI: the annotation table, byte for byte, against the files the unmodified reference wrote
(every golden: texts with and without N, threads 1 and 2) and against the oracle's restatement in both of its modes
(with the reference's reuse of the last k-mer's search, the default; every k-mer searched afresh); the number of
k-mers searched and of extension steps; chunked runs (the reuse's state crossing chunks); the reference's argument checks."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from oracle import pyoracle as po
from tests import common
from tests.test_oracle_ann import CASES, WITH_N, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,k,tau,threads", CASES)
def test_ann_table_matches_reference_and_oracle(workdir, name, k, tau, threads):
    """Every golden, texts with N included, threads 1 and 2: the table is the file the unmodified reference wrote, byte for
    byte (the default: with the reference's reuse of the last k-mer's search), and k-mers and extension steps are the
    restatement's; with every k-mer searched afresh (ann_trail_reuse = 0) table and steps are the restatement's in that mode."""
    base = common.build_golden_index(workdir, name)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    got, st = ctx.ann_build(k, tau, threads)
    assert np.array_equal(got, load_golden(name, k, tau, threads))
    want, ost = orc.ann_build(k, tau, threads, trail_reuse=True)
    assert np.array_equal(got, want)
    assert st["queries"] == ost["queries"] and st["n_ext"] == ost["n_ext"]
    got, st = ctx.ann_build(k, tau, threads, trail_reuse=False)
    want, ost = orc.ann_build(k, tau, threads, trail_reuse=False)
    assert np.array_equal(got, want)
    assert st["queries"] == ost["queries"] and st["n_ext"] == ost["n_ext"]
    if name not in WITH_N:
        assert np.array_equal(got, load_golden(name, k, tau, threads))
    ctx.close()


def test_ann_in_chunks_and_on_a_synthetic_text(workdir, monkeypatch):
    base, seqs = common.synth_index(workdir, "ann60k", 60_000, seed=41, n_seqs=3, repeat_rich=True)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    for k, tau in ((22, 1), (18, 2)):
        want, ost = orc.ann_build(k, tau, 1, trail_reuse=True)
        got, st = ctx.ann_build(k, tau)
        assert np.array_equal(got, want) and st["n_ext"] == ost["n_ext"]
        monkeypatch.setenv("GPUMAP_ANN_CHUNK", "5000")
        monkeypatch.setenv("GPUMAP_ANN_PIECE", "4096")            # the suffix array in pieces (texts of 2^31 symbols and more)
        got2, st2 = ctx.ann_build(k, tau)
        monkeypatch.delenv("GPUMAP_ANN_CHUNK")
        monkeypatch.delenv("GPUMAP_ANN_PIECE")
        assert np.array_equal(got2, want) and st2["queries"] == st["queries"]
        assert np.count_nonzero(want) > 100
    # gpumap_ann_file_write: the file the reference's index_read would pick up
    path = os.path.join(workdir, "ann60k.ann.18.2")
    assert ma.lib().gpumap_ann_file_write(path.encode(), got.ctypes.data, len(got), 18, 2) == 0
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw[:32].view(np.int64).tolist() == [0x0fcb0fcb00050001, 18, 2, len(got)] and np.array_equal(raw[32:], got)
    ctx.close()


def test_ann_reuse_next_to_n(workdir, monkeypatch):
    """Texts in which every few k-mers hold an N: the reference's tree then keeps hits of other letters and other budgets
    (src/blocksearch.c:41-43, :63-66, :73-86, :156-157) and its table depends on the order of the k-mers.  The table and
    the steps are the restatement's with the reuse (pinned to the reference's files in tests/test_oracle_ann.py), whole and
    in chunks of a few hundred k-mers (the state that crosses chunks), for one and for several threads' job lists."""
    base = common.build_golden_index(workdir, "annN")         # tools/make_golden_ann.py: N in every few k-mers
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    split = 0
    for k, tau, threads in ((8, 3, 1), (9, 2, 1), (12, 3, 1), (13, 2, 3), (16, 1, 1), (7, 3, 6), (25, 2, 1)):
        want, ost = orc.ann_build(k, tau, threads, trail_reuse=True)
        fresh, _ = orc.ann_build(k, tau, threads, trail_reuse=False)
        got, st = ctx.ann_build(k, tau, threads)
        assert np.array_equal(got, want), (k, tau, threads, int(np.count_nonzero(got != want)), int(np.count_nonzero(fresh != want)))
        assert st["queries"] == ost["queries"] and st["n_ext"] == ost["n_ext"]
        monkeypatch.setenv("GPUMAP_ANN_CHUNK", "300")
        got2, st2 = ctx.ann_build(k, tau, threads)
        monkeypatch.delenv("GPUMAP_ANN_CHUNK")
        assert np.array_equal(got2, want) and st2["n_ext"] == ost["n_ext"]
        split += st["split"]
        got3, st3 = ctx.ann_build(k, tau, threads, trail_reuse=False)
        assert np.array_equal(got3, fresh)
    assert split > 0                                       # some k-mers were searched as two queries
    ctx.close()


def test_ann_bad_arguments(workdir):
    ctx = ma.Context.from_files(common.build_golden_index(workdir, "repeats"), 0)
    for k, tau in ((1, 1), (-1, 1), (10, 0), (20, -1), (3, 4), (25, 8), (3, 3)):          # test/test_index_ann.c:39-44
        with pytest.raises(ma.GpumapError):
            ctx.ann_build(k, tau)
    ctx.close()


def test_second_jump_table_changes_nothing(workdir, monkeypatch):
    """Where the context's jump table is longer than the left block of a k-mer, gpumap_ann_build builds a second one of the
    block's length (gm_ctx_build_jump2) that stands in for the block's exact steps in k_search<true> and for prefixes of
    the path of k_sc_prepare: the same table, k-mers and extension steps as without it; and reads whose leaf blocks only
    the second table fits map as the oracle says (located hits and extension steps) while it is in place."""
    from mapper_amd import synth
    base, seqs = common.synth_index(workdir, "ann2m", 2_200_000, seed=43, n_seqs=2, repeat_rich=True)
    ctx = ma.Context.from_files(base, 0)
    ctx.set_option("jump_k", 12)                       # longer than the 12 or 13 letters of a 25-mer's left block allow
    monkeypatch.setenv("GPUMAP_ANN_JUMP2", "0")
    want, st0 = ctx.ann_build(25, 1)
    monkeypatch.delenv("GPUMAP_ANN_JUMP2")
    got, st1 = ctx.ann_build(25, 1)                    # builds the table of 11 letters (4^11 <= the text's length)
    assert np.array_equal(got, want)
    assert st0["jump2_k"] == 0 and st1["jump2_k"] == 11
    assert (st0["queries"], st0["n_ext"], st0["hits"]) == (st1["queries"], st1["n_ext"], st1["hits"])
    assert np.count_nonzero(got) > 1000
    codes, offs = synth.make_reads(seqs, 3000, 36, 2, seed=44, n_frac=0.01)
    d = po.Index(base).map_batch(codes.reshape(-1), offs, 2, threads=8)
    h = ctx.map_batch(codes.reshape(-1), offs, 2)
    assert np.array_equal(h.located(), d.located())
    assert h.n_ext == d.n_ext
    ctx.close()
