"""mapper() end to end (src/mapper.c:3-47 with the GPU batch in place of the stub's print): a read file
in, one line per located hit out.  Expected lines come from the oracle's hits formatted with the
host's txt_pos_to_str (checked against the reference's KATs in test_oracle_golden.py)."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from oracle import pyoracle as po
from tests import common

pytestmark = pytest.mark.gpu

LETTERS = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _write(path, fmt, tags, codes, offs):
    with open(path, "wb") as f:
        for i, t in enumerate(tags):
            s = LETTERS[codes[offs[i]:offs[i + 1]]].tobytes()
            if fmt == "fastq":
                f.write(b"@" + t.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
            elif fmt == "fasta":
                f.write(b">" + t.encode() + b"\n" + s + b"\n")
            else:
                f.write(s + b"\n")


def _expected(index, orc, tags, codes, offs, tau):
    d = orc.map_batch(codes, offs, tau, threads=8)
    return ["%s\t%s\t%d" % (tags[r], index.pos_to_str(int(p)), s) for r, p, s in d.located()]


@pytest.mark.parametrize("fmt", ["fastq", "fasta", "raw"])
@pytest.mark.parametrize("chunk_bytes", [None, 20000])
def test_mapper_lines_match_oracle(workdir, fmt, chunk_bytes, monkeypatch):
    base, seqs = common.synth_index(workdir, "rnd300k", 300_000, seed=21, n_seqs=3)
    codes, offs = synth.make_reads(seqs, 3000, 100, 2, seed=77, n_frac=0.01)
    codes = codes.reshape(-1)
    n = len(offs) - 1
    tags = ["read_%d/1 len=100" % i for i in range(n)] if fmt != "raw" else [str(i + 1) for i in range(n)]
    qfile = os.path.join(workdir, "reads_%s.%s" % (chunk_bytes, fmt))
    _write(qfile, fmt, tags, codes, offs)
    if chunk_bytes:
        monkeypatch.setenv("GPUMAP_CHUNK_BYTES", str(chunk_bytes))     # several chunks (default: 256 MB, src/io.h:13)
    index = ma.HostIndex.read(base)
    ctx = ma.Context.from_index(index, 0)
    out = qfile + ".out"
    ctx.mapper(qfile, index, 2, out)
    got = open(out).read().split("\n")[:-1]
    want = _expected(index, po.Index(base), tags, codes, offs, 2)
    assert len(want) >= n * 0.99
    assert got == want
    ctx.close()
    index.close()


def test_mapper_ragged_reads_and_errors(workdir):
    base, seqs = common.synth_index(workdir, "rnd300k", 300_000, seed=21, n_seqs=3)
    rng = np.random.default_rng(5)
    reads, tags = [], []
    g = np.concatenate(seqs)
    for i, m in enumerate([20, 31, 64, 65, 100, 128, 129, 150, 192, 193, 255, 256, 300, 0, 12]):
        p = int(rng.integers(0, len(seqs[0]) - 400))
        reads.append(seqs[0][p:p + m].copy())
        tags.append("r%d" % i)
    codes = np.concatenate(reads).astype(np.uint8)
    offs = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.int64)
    qfile = os.path.join(workdir, "ragged.fastq")
    _write(qfile, "fastq", tags, codes, offs)
    index = ma.HostIndex.read(base)
    ctx = ma.Context.from_index(index, 0)
    ctx.mapper(qfile, index, 2, qfile + ".out")
    got = open(qfile + ".out").read().split("\n")[:-1]
    assert got == _expected(index, po.Index(base), tags, codes, offs, 2)
    # a quality line of another length than the sequence fails the whole call (src/seqread.c:47-49)
    bad = os.path.join(workdir, "bad.fastq")
    with open(bad, "w") as f:
        f.write("@a\nACGT\n+\nIII\n")
    with pytest.raises(ma.GpumapError):
        ctx.mapper(bad, index, 2, bad + ".out")
    with pytest.raises(ma.GpumapError):
        ctx.mapper(os.path.join(workdir, "missing.fastq"), index, 2, bad + ".out")
    ctx.close()
    index.close()
