"""Read sharding across GPUs: one process per GPU, each with a full index replica.

Reads are independent, so there is no collective on the data path: rank g maps the contiguous range
[g*R/G, (g+1)*R/G) of the batch and the per-rank hit lists are concatenated on rank 0 in rank order,
which is input order.  The only collectives are the one-off index broadcast (bench.py / replicate())
and this final gather of results.
"""
import numpy as np


def shard_bounds(n_reads, world):
    return [n_reads * g // world for g in range(world + 1)]


def shard_of(codes, offsets, rank, world):
    """(codes, offsets, first_read) of this rank's contiguous share of a batch."""
    offsets = np.asarray(offsets, dtype=np.int64)
    b = shard_bounds(len(offsets) - 1, world)
    lo, hi = b[rank], b[rank + 1]
    return np.asarray(codes)[offsets[lo]:offsets[hi]], offsets[lo:hi + 1] - offsets[lo], lo


def map_sharded(map_fn, codes, offsets, tau, rank, world, dist=None, dst=0):
    """map_fn(codes, offsets, tau) -> int64 [k][3] located hits (read, text_pos, score) of a batch.
    Returns, on rank dst, the located hits of the whole batch in input order; None elsewhere."""
    c, o, first = shard_of(codes, offsets, rank, world)
    local = np.array(map_fn(c, o, tau), dtype=np.int64, copy=True).reshape(-1, 3)
    local[:, 0] += first
    if world == 1:
        return local
    parts = [None] * world if rank == dst else None
    dist.gather_object(local, parts, dst=dst)
    if rank != dst:
        return None
    return np.concatenate(parts, axis=0)
