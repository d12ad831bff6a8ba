"""Large indexes (built on the device) checked through properties that do not need a CPU search of the
whole batch: every read drawn with at most tau substitutions must be located at its origin, on its
strand (completeness of the pigeonhole schedule); plus bit-exact parity with the
oracle on a sample, the oracle reading the reference-layout arrays of the same device build.
The second case has a text longer than 2^32 symbols (33-bit suffix-array entries, 64-bit intervals)."""
import numpy as np
import pytest
import torch

import mapper_amd as ma
from mapper_amd import synth_device
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _origin_check(h, origin, strand, lens, m):
    """Expected (loc_fwd, loc_strand) of each read's origin (src/index_txt.c:492-496): forward strand: text
    position of the window's first base; reverse: of its last base.  Text position = position in the
    concatenated sequences + index of the sequence (one separator behind every earlier sequence)."""
    bounds = np.cumsum(lens)
    first = origin + np.searchsorted(bounds, origin, side="right")
    last = origin + m - 1 + np.searchsorted(bounds, origin + m - 1, side="right")
    want = np.where(strand == 1, last, first)
    key_have = set(zip(h.loc_read.tolist(), h.loc_fwd.tolist(), h.loc_strand.tolist()))
    missing = [r for r in range(len(origin)) if (r, int(want[r]), int(strand[r])) not in key_have]
    return missing


@pytest.mark.parametrize("total_len,n_seqs,n_reads,sample,shapes", [
    (300_000_000, 3, 100_000, 20_000, [(100, 2)]),
    (2_200_000_000, 5, 100_000, 5_000, [(100, 2)]),
    # BASELINE configs 4 and 5 at their full index size (n = 6 200 000 048: the bench's own index)
    (3_100_000_000, 24, 200_000, 6_000, [(100, 2), (150, 4)])])
def test_large_device_built_index(total_len, n_seqs, n_reads, sample, shapes):
    g, lens, _ = synth_device.make_genome(total_len, 4 if n_seqs == 24 else 9, n_seqs=n_seqs)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ctx = ma.Context.build_device(lens, 0, keep_reference=True, device_ptr=g.data_ptr())
    n = 2 * (total_len + n_seqs)
    batches = []
    for m, tau in shapes:
        reads, origin, strand = synth_device.make_reads(g, lens, n_reads, m, tau, seed=31 + m, return_origin=True)
        batches.append((m, tau, reads.cpu().numpy(), origin.cpu().numpy(), strand.cpu().numpy()))
        del reads
    del g
    torch.cuda.empty_cache()
    orc = None
    for m, tau, codes, origin, strand in batches:
        offs = np.arange(n_reads + 1, dtype=np.int64) * m
        h = ctx.map_batch(codes.reshape(-1), offs, tau)
        assert (h.status == 0).all()
        assert (h.loc_score <= tau).all()
        assert np.array_equal(h.loc_strand, (h.loc_pos >= n // 2).astype(np.uint8))
        assert _origin_check(h, origin, strand, lens, m) == []
        # sorted output: reads in input order, positions ascending inside a read
        key = h.loc_read.astype(np.float64) * float(n) + h.loc_pos
        assert (np.diff(key) > 0).all()
        # the pipeline's compact results on the same batch: same keys
        ctx.set_option("compact_results", 1)
        b = ctx.upload(codes.reshape(-1), offs)
        b.run(tau)
        hc = b.fetch()
        b.close()
        ctx.set_option("compact_results", 0)
        assert np.array_equal(hc.located(), h.located()) and np.array_equal(hc.read_first_loc, h.read_first_loc)
        # sample against the oracle on the same index
        if orc is None:
            ref = ctx.reference_arrays(want_text=False)
            orc = po.Index.from_arrays(ref["occ"], ref["c"], ref["sa"], ref["sar_bits"], ref["txt_len"], ref["wil_cnt"])
        d = orc.map_batch(codes[:sample].reshape(-1), offs[:sample + 1], tau, threads=8)
        hs = ctx.map_batch(codes[:sample].reshape(-1), offs[:sample + 1], tau, want_intervals=True)
        assert np.array_equal(hs.located(), d.located())
        assert np.array_equal(hs.intervals(), d.intervals())
        assert hs.n_ext == d.n_ext
        hv = ctx.map_batch(codes[:sample].reshape(-1), offs[:sample + 1], tau)         # the k_verify path (what the bench times)
        assert np.array_equal(hv.located(), d.located()) and hv.n_ext == d.n_ext
    ctx.close()
