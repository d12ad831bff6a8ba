"""gpumap_query_all_noalloc (the entry point that allocates nothing: its staging is a static device array of
the library) against gpumap_query_all, through the C-ABI, bit for bit; two calls in a row, the second with
a smaller input, so that what the first call left in the shared array cannot leak into the second."""
import numpy as np
import pytest

import mapper_amd as ma
from tests import common

pytestmark = pytest.mark.gpu


def _intervals(ctx, n_txt, count, seed):
    """`count` bi-intervals reached by random walks of gpumap_query_all from the root interval."""
    rng = np.random.default_rng(seed)
    fp, rp, sz = [0], [0], [n_txt]
    cur = np.array([[0, 0, n_txt]], dtype=np.int64)
    while len(fp) < count:
        end = int(rng.integers(0, 2))
        ch = ctx.query_all(end, cur[:, 0], cur[:, 1], cur[:, 2]).reshape(-1, 3)
        ch = ch[ch[:, 2] > 0]
        ch = ch[rng.permutation(len(ch))[:4000]]
        if len(ch) == 0:
            cur = np.array([[0, 0, n_txt]], dtype=np.int64)
            continue
        cur = ch
        fp += ch[:, 0].tolist(); rp += ch[:, 1].tolist(); sz += ch[:, 2].tolist()
    return (np.array(x[:count], dtype=np.int64) for x in (fp, rp, sz))


def test_query_all_noalloc_equals_query_all(workdir):
    base = common.build_golden_index(workdir, "small")
    ctx = ma.Context.from_files(base, 0)
    n_txt = ma.HostIndex.read(base).txt_len
    fp, rp, sz = _intervals(ctx, n_txt, ma.NOALLOC_MAX_QUERIES, seed=5)
    # first call: the maximum size; second call: fewer intervals, other values, the other end
    import torch
    for count, end, lo in ((ma.NOALLOC_MAX_QUERIES, ma.QUERY_PREFIX, 0), (777, ma.QUERY_SUFFIX, 1234), (1, ma.QUERY_PREFIX, 9)):
        a, b, c = fp[lo:lo + count], rp[lo:lo + count], sz[lo:lo + count]
        want = ctx.query_all(end, a, b, c)              # (first: this one grows the context's staging with hipMalloc)
        free_before = torch.cuda.mem_get_info(0)[0]
        got = ctx.query_all_noalloc(end, a, b, c)
        assert torch.cuda.mem_get_info(0)[0] == free_before      # the property itself: the device heap did not change
        assert got.shape == (count, 5, 3)
        assert np.array_equal(got, want)
    # invalid intervals give -1 rows exactly as gpumap_query_all does; an empty call succeeds
    bad = np.array([-1, n_txt, 3], dtype=np.int64)
    assert np.array_equal(ctx.query_all_noalloc(0, bad, bad, bad), ctx.query_all(0, bad, bad, bad))
    assert ctx.query_all_noalloc(0, [], [], []).shape == (0, 5, 3)
    # beyond the maximum: an error, not a truncated result
    big = np.zeros(ma.NOALLOC_MAX_QUERIES + 1, dtype=np.int64)
    with pytest.raises(ma.GpumapError):
        ctx.query_all_noalloc(0, big, big, big)
    # a second context on the same GPU shares the array (same device address): still the same answers
    ctx2 = ma.Context.from_files(base, 0)
    assert np.array_equal(ctx2.query_all_noalloc(1, fp[:500], rp[:500], sz[:500]), ctx.query_all(1, fp[:500], rp[:500], sz[:500]))
    ctx2.close()
    ctx.close()
