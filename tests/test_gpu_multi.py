"""Several GPUs behind the C-ABI (gpumap_multi) on a box with one GPU: G members share the device ("test
without 8 GPUs: logical sharding on 1 device must give byte-identical concatenated output for G in {1,2,4,8}"), and
the RCCL path (dlopen of librccl, ncclCommInitAll, grouped ncclBroadcast of the four arrays) is driven on a
communicator of one rank.  Results are compared with one context's, which test_gpu_parity.py compares with the oracle."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(workdir):
    base, seqs = common.synth_index(workdir, "rep400k", 400_000, seed=31, n_seqs=2, repeat_rich=True)
    hix = ma.HostIndex.read(base)
    ctx = ma.Context.from_index(hix, 0)
    codes, offs = synth.make_reads(seqs, 9001, 100, 2, seed=77, n_frac=0.01)
    want = ctx.map_batch(codes.reshape(-1), offs, 2, want_intervals=True)
    yield base, hix, ctx, codes.reshape(-1), offs, want
    ctx.close()


@pytest.mark.parametrize("g", [1, 2, 4, 8])
def test_members_on_one_device_give_the_output_of_one_gpu(setup, g):
    base, hix, ctx, codes, offs, want = setup
    m = ma.Multi.from_index(hix, g, devices=[0] * g)
    assert len(m) == g and not m.used_rccl
    for intervals in (True, False):
        h = m.map_batch(codes, offs, 2, want_intervals=intervals)
        assert np.array_equal(h.status, want.status)
        assert np.array_equal(h.read_first_loc, want.read_first_loc)
        for k in ("loc_read", "loc_pos", "loc_fwd", "loc_seq", "loc_seqpos", "loc_strand", "loc_score"):
            assert np.array_equal(getattr(h, k), getattr(want, k)), k
        assert h.n_ext == want.n_ext
        if intervals:
            assert np.array_equal(h.intervals(), want.intervals())
    # fewer reads than members, and none
    h = m.map_batch(codes[:300], offs[:4], 2)
    assert np.array_equal(h.located(), ctx.map_batch(codes[:300], offs[:4], 2).located())
    assert m.map_batch(np.zeros(0, np.uint8), np.zeros(1, np.int64), 2).n_locs == 0
    m.close()


def test_a_member_with_compact_results_changes_nothing(setup):
    """gpumap_multi_ctx hands the members out and options only change how results travel: a member set to
    "compact_results" returns keys, which gpumap_multi_map_batch expands before it concatenates."""
    base, hix, ctx, codes, offs, want = setup
    m = ma.Multi.from_index(hix, 3, devices=[0, 0, 0])
    mem = m.members()
    mem[1].set_option("compact_results", 1)
    mem[2].set_option("pinned_results", 1)
    mem[2].set_option("compact_results", 1)
    h = m.map_batch(codes, offs, 2)
    assert np.array_equal(h.read_first_loc, want.read_first_loc)
    for k in ("loc_read", "loc_pos", "loc_fwd", "loc_seq", "loc_seqpos", "loc_strand", "loc_score"):
        assert np.array_equal(getattr(h, k), getattr(want, k)), k
    m.close()


def test_rccl_broadcast_path_on_one_rank(setup, monkeypatch):
    base, hix, ctx, codes, offs, want = setup
    monkeypatch.setenv("GPUMAP_MULTI_FORCE_RCCL", "1")
    m = ma.Multi.from_ctx(ctx, 2, devices=[0, 0])
    assert m.used_rccl                      # librccl loaded, communicator made, the four arrays broadcast (in place)
    h = m.map_batch(codes, offs, 2)
    assert np.array_equal(h.located(), want.located()) and h.n_ext == want.n_ext
    m.close()
    # the root context is the caller's and still works
    assert np.array_equal(ctx.map_batch(codes, offs, 2).located(), want.located())


def test_mapper_over_members_writes_what_one_gpu_writes(setup, workdir, monkeypatch):
    base, hix, ctx, codes, offs, want = setup
    fq = os.path.join(workdir, "multi.fq")
    with open(fq, "w") as f:
        for r in range(len(offs) - 1):
            s = synth.codes_to_text(codes[offs[r]:offs[r + 1]]).decode()
            f.write("@r%d\n%s\n+\n%s\n" % (r, s, "I" * len(s)))
    monkeypatch.setenv("GPUMAP_CHUNK_BYTES", "150000")          # many chunks going round the members
    one = os.path.join(workdir, "multi_one.txt")
    ctx.mapper(fq, hix, 2, one)
    m = ma.Multi.from_ctx(ctx, 3, devices=[0, 0, 0])
    out = os.path.join(workdir, "multi_three.txt")
    m.mapper(fq, hix, 2, out)
    assert open(out, "rb").read() == open(one, "rb").read()
    assert os.path.getsize(one) > 100000
    m.close()


def test_bench_runs_its_distributed_path_on_one_gpu():
    """bench.py --force-dist: what the driver's N > 1 runs execute (process group on RCCL, broadcast of the index arrays
    through __cuda_array_interface__ views in 1-GiB pieces, gpumap_create_from_device on the received buffers, barrier and
    max over ranks, replica check across ranks and against the oracle), with a world of one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "ecoli", "--reads", "20000", "--steps", "3", "--warmup", "1",
                        "--force-dist", "--no-pcie", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines[:5]                       # one JSON line and nothing else (RCCL's banner goes to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["replica_check"]["identical_across_ranks"] and line["replica_check"]["identical_to_oracle"]
    assert line["setup_s"]["rccl_broadcast"] > 0 and line["setup_s"]["replica_from_device"] > 0
    assert line["cpu_baseline"]["parity_on_sample"]["identical"]
