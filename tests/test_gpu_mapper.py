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


@pytest.mark.parametrize("fmt", ["fastq", "fasta", "raw"])
def test_device_parsing_and_formatting_equal_the_host_path(workdir, fmt, monkeypatch):
    """gpumap_mapper parses the chunks and formats the lines on the device (gm_parse.hpp); with GPUMAP_MAPPER_HOST_PARSE=1
    the host does both (gm_reads.cpp, checked against the line-by-line model of the reference's reader in
    tests/test_host.py).  Files with what the reader's rules are about: NULs inside tags and sequences (a line's content
    ends there), CRs, empty lines and empty reads, tags of every length, a last record that is incomplete, a last line
    without newline; cut into chunks of several sizes.  Both paths write the same bytes."""
    base, seqs = common.synth_index(workdir, "rnd300k", 300_000, seed=21, n_seqs=3)
    codes, offs = synth.make_reads(seqs, 4000, 100, 2, seed=78, n_frac=0.01)
    codes = codes.reshape(-1)
    rng = np.random.default_rng(9)
    recs = []
    for i in range(len(offs) - 1):
        m = 100 if rng.random() < 0.8 else int(rng.integers(0, 101))
        s = LETTERS[codes[offs[i]:offs[i] + m]].tobytes()
        u = rng.random()
        if u < 0.03 and m > 40:
            s = s[:40] + b"\0" + s[41:]                       # the read ends at the NUL
        elif u < 0.06:
            s = s + b"\r"                                     # a CR is a letter like any other unknown one
        tag = b"r%d" % i + b" x" * int(rng.integers(0, 30))
        if rng.random() < 0.02:
            tag = tag[:2] + b"\0" + tag[2:]
        if rng.random() < 0.01:
            tag = b""
        if fmt == "fastq":
            recs += [b"@" + tag if rng.random() < 0.99 else b"", s, b"+" + (b"anything" if rng.random() < 0.1 else b""), b"I" * len(s.split(b"\0")[0])]
        elif fmt == "fasta":
            recs += [b">" + tag, s]
        else:
            recs += [s if i or not s.startswith((b"@", b">")) else b"A" + s]
    for variant, data in (("whole", b"\n".join(recs) + b"\n"), ("open", b"\n".join(recs)), ("cut", b"\n".join(recs[:-1]) + b"\n")):
        qfile = os.path.join(workdir, "tricky_%s.%s" % (variant, fmt))
        open(qfile, "wb").write(data)
        index = ma.HostIndex.read(base)
        ctx = ma.Context.from_index(index, 0)
        for chunk_bytes in ("1", "777", "50000", "100000000"):
            monkeypatch.setenv("GPUMAP_CHUNK_BYTES", chunk_bytes)
            monkeypatch.setenv("GPUMAP_MAPPER_HOST_PARSE", "1")
            ctx.mapper(qfile, index, 2, qfile + ".host")
            monkeypatch.setenv("GPUMAP_MAPPER_HOST_PARSE", "0")
            ctx.mapper(qfile, index, 2, qfile + ".dev")
            a, b = open(qfile + ".host", "rb").read(), open(qfile + ".dev", "rb").read()
            assert a == b, (variant, chunk_bytes, len(a), len(b))
            assert a.count(b"\n") > 2500
            if chunk_bytes == "1" and variant == "whole":
                break                                        # (a chunk per record: once is enough)
        ctx.close()
        index.close()


def test_device_path_reports_a_bad_quality_line_in_a_later_chunk(workdir, monkeypatch):
    base, seqs = common.synth_index(workdir, "rnd300k", 300_000, seed=21, n_seqs=3)
    index = ma.HostIndex.read(base)
    ctx = ma.Context.from_index(index, 0)
    bad = os.path.join(workdir, "bad_later.fastq")
    with open(bad, "w") as f:
        for i in range(200):
            f.write("@r%d\nACGTACGTACGTACGTACGTACGTACGT\n+\n%s\n" % (i, "I" * (28 if i != 150 else 27)))
    monkeypatch.setenv("GPUMAP_CHUNK_BYTES", "2000")
    with pytest.raises(ma.GpumapError):
        ctx.mapper(bad, index, 2, bad + ".out")
    ctx.close()
    index.close()
