"""GPU parity of gpumap_ann_build (ann_build, src/index_ann.c:56-257; SURVEY 8f rank 4: the reference's consumer of
blocksc_trail) through the C-ABI: the annotation table, byte for byte, against the files the unmodified reference wrote
(texts without N: there the reference's reuse of searches is exact) and against the oracle's restatement with every
k-mer searched afresh (all texts; tests/test_oracle_ann.py pins the restatement to the reference); the number of
k-mers searched and of extension steps; chunked runs; the reference's argument checks."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from oracle import pyoracle as po
from tests import common
from tests.test_oracle_ann import CASES, WITH_N, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,k,tau,threads", [c for c in CASES if c[3] == 1])
def test_ann_table_matches_reference_and_oracle(workdir, name, k, tau, threads):
    base = common.build_golden_index(workdir, name)
    ctx = ma.Context.from_files(base, 0)
    got, st = ctx.ann_build(k, tau)
    want, ost = po.Index(base).ann_build(k, tau, 1, trail_reuse=False)
    assert np.array_equal(got, want)
    assert st["queries"] == ost["queries"] and st["n_ext"] == ost["n_ext"]
    if name not in WITH_N:
        assert np.array_equal(got, load_golden(name, k, tau, 1))
    ctx.close()


def test_ann_in_chunks_and_on_a_synthetic_text(workdir, monkeypatch):
    base, seqs = common.synth_index(workdir, "ann60k", 60_000, seed=41, n_seqs=3, repeat_rich=True)
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    for k, tau in ((22, 1), (18, 2)):
        want, ost = orc.ann_build(k, tau, 1, trail_reuse=False)
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


def test_ann_bad_arguments(workdir):
    ctx = ma.Context.from_files(common.build_golden_index(workdir, "repeats"), 0)
    for k, tau in ((1, 1), (-1, 1), (10, 0), (20, -1), (3, 4), (25, 8), (3, 3)):          # test/test_index_ann.c:39-44
        with pytest.raises(ma.GpumapError):
            ctx.ann_build(k, tau)
    ctx.close()
