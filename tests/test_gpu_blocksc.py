"""GPU parity of gpumap_blocksc_batch (blocksc_trail + scsearch_fw, src/blocksearch.c:7-103, :314-378)
through the C-ABI: against the dumps of the compiled reference (tests/golden/*.sc.*.bin) and, on larger random
batches, against the oracle.  Bit-exact: status, interval hits (fp, rp, sz, dp, score with the N count added,
mismatch bitmap), located positions, and the count of bwt_query_all calls."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from oracle import pyoracle as po
from tests import common
from tests.test_oracle_blocksc import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("index,queries,dump,tau", CASES)
def test_blocksc_matches_reference_dump(workdir, index, queries, dump, tau):
    base = common.build_golden_index(workdir, index)
    ctx = ma.Context.from_files(base, 0)
    codes, offs = common.encode_reads(common.load_reads(queries))
    ref = po.HitDump.load(os.path.join(common.GOLDEN, dump))
    want_ext = po.Index(base).blocksc_batch(codes, offs, tau, threads=2).n_ext
    for jump_k in (-1, 0):
        ctx.set_option("jump_k", jump_k)
        h = ctx.blocksc_batch(codes, offs, tau)
        assert np.array_equal(h.status, ref.status)
        assert np.array_equal(h.intervals(), ref.intervals())
        assert np.array_equal(h.located(), ref.located())
        assert h.n_ext == want_ext
    ctx.close()


@pytest.mark.parametrize("k,tau", [(25, 1), (25, 2), (32, 3), (31, 4), (16, 2), (64, 5)])
def test_blocksc_random_kmers_match_oracle(workdir, k, tau):
    """k-mers drawn from a repeat-rich text (both strands, some with substitutions so that they are not in the text,
    some with N), as ann_build_mt would query them, one length per batch."""
    base, seqs = common.synth_index(workdir, "rep400k", 400_000, seed=31, n_seqs=2, repeat_rich=True)
    codes, offs = synth.make_reads(seqs, 6000, k, 1, seed=900 + k, n_frac=0.03)
    ctx = ma.Context.from_files(base, 0)
    d = po.Index(base).blocksc_batch(codes.reshape(-1), offs, tau, threads=8)
    h = ctx.blocksc_batch(codes.reshape(-1), offs, tau)
    assert np.array_equal(h.status, d.status)
    assert np.array_equal(h.intervals(), d.intervals())
    assert np.array_equal(h.located(), d.located())
    assert h.n_ext == d.n_ext
    assert h.n_hits > 0
    ctx.close()


def test_blocksc_mixed_lengths_and_degenerate_queries(workdir):
    base = common.build_golden_index(workdir, "dense")
    ctx = ma.Context.from_files(base, 0)
    orc = po.Index(base)
    rng = np.random.default_rng(3)
    text = "".join(l.strip() for l in open(os.path.join(common.GOLDEN, "dense.fa")) if not l.startswith(">"))
    qs = []
    for m in (1, 2, 3, 4, 5, 7, 8, 11, 12, 13, 20):
        for _ in range(40):
            a = int(rng.integers(0, len(text) - m))
            qs.append(text[a:a + m])
    qs += ["", "T", "A", "N", "NN", "ANA", "TTTTTTTTTTTT", "ACGT" * 64, "ACGT" * 64 + "A"]
    codes, offs = common.encode_reads(qs)
    for tau in (0, 1, 2, 3):
        d = orc.blocksc_batch(codes, offs, tau, threads=4)
        h = ctx.blocksc_batch(codes, offs, tau)
        assert np.array_equal(h.status, d.status), tau
        assert np.array_equal(h.intervals(), d.intervals()), tau
        assert np.array_equal(h.located(), d.located()), tau
        assert h.n_ext == d.n_ext
    # more distinct lengths than one batch can hold rows for: an error, not a wrong answer
    many = [text[:m] for m in range(1, 60)]
    codes, offs = common.encode_reads(many)
    with pytest.raises(ma.GpumapError):
        ctx.blocksc_batch(codes, offs, 3)
    ctx.close()
