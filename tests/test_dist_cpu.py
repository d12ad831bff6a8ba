"""The N > 1 path on CPU: world_size-2 gloo processes shard a batch, map their shares and gather the
hits in input order.  The compute stand-in here is the oracle (test infrastructure); on the GPU box
the same helper is driven with Context.map_batch (tests/test_gpu_parity.py covers the logical shards)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from mapper_amd import distributed as gd
from mapper_amd import synth
from tests import common


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, base, codes, offs, out_path):
    import torch.distributed as dist
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = po.Index(base)
    res = gd.map_sharded(lambda c, o, t: orc.map_batch(c, o, t, threads=1).located(), codes, offs, 2, rank, world, dist)
    if rank == 0:
        np.save(out_path, res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 1000):
        for g in (1, 2, 3, 8):
            b = gd.shard_bounds(n, g)
            assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))


def test_gloo_two_ranks_concatenate_in_input_order(workdir):
    from oracle import pyoracle as po
    base, seqs = common.synth_index(workdir, "dist50k", 50_000, seed=41, n_seqs=2)
    codes, offs = synth.make_reads(seqs, 301, 64, 2, seed=42)
    codes = codes.reshape(-1)
    whole = po.Index(base).map_batch(codes, offs, 2, threads=2).located()
    out = os.path.join(workdir, "dist_out.npy")
    mp.spawn(_worker, args=(2, _free_port(), base, codes, offs, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), whole)
