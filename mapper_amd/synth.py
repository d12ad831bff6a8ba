"""Deterministic synthetic references and reads.

No genome files exist where this runs, so every "E. coli / chr21 / GRCh38" configuration is a
synthetic stand-in of the same size, generated from a seed with a counter-based splitmix64 (the
value at index i depends only on (seed, stream, i), so chunked and whole generation agree).
Symbols are the reference's codes: A=0 C=1 G=2 T=3 N=4 (src/index_sym.c:11-13).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, idx):
    """splitmix64 output number idx (array of counters) of the generator seeded with `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class Stream:
    """Sequential view of one splitmix64 stream."""

    def __init__(self, seed, stream=0):
        self.seed = int(splitmix64(seed, np.array([stream], dtype=np.uint64))[0])
        self.pos = 0

    def u64(self, n):
        out = splitmix64(self.seed, np.arange(self.pos, self.pos + n, dtype=np.uint64))
        self.pos += n
        return out

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def integers(self, lo, hi, n):
        """n integers in [lo, hi) (multiply-shift; bias < 2^-32 irrelevant here)."""
        span = np.uint64(hi - lo)
        return (lo + (((self.u64(n) >> np.uint64(32)) * span) >> np.uint64(32)).astype(np.int64)).astype(np.int64)


def random_bases(stream, n, gc=0.5):
    u = stream.uniform(n)
    at = (1.0 - gc) / 2.0
    out = np.zeros(n, dtype=np.uint8)                  # A
    out[u >= at] = 1                                   # C
    out[u >= at + gc / 2.0] = 2                        # G
    out[u >= at + gc] = 3                              # T
    return out


def _mutate(stream, seq, rate):
    """Substitute each base with probability `rate` by a different base."""
    n = len(seq)
    hit = stream.uniform(n) < rate
    shift = stream.integers(1, 4, n).astype(np.uint8)
    out = seq.copy()
    out[hit] = (seq[hit] + shift[hit]) & 3
    return out


def make_genome(total_len, seed, n_seqs=1, gc=0.5, repeat_rich=False):
    """Returns (list of code arrays, list of names).  Lengths sum to total_len.

    repeat_rich: i.i.d. background + 20 interspersed families (300 bp, 5-15 % per-copy divergence,
    about 10 % of the sequence) + tandem arrays (171-bp monomer, 2 % divergence, about 2 %) +
    N runs (one long, several short), so that repeats, text N and deep backtracking occur.
    """
    st = Stream(seed, 0)
    g = random_bases(st, total_len, gc)
    if repeat_rich:
        rs = Stream(seed, 1)
        n_fam = 20
        fam_len = 300
        copies_total = max(n_fam, int(0.10 * total_len / fam_len))
        cons = [random_bases(rs, fam_len, gc) for _ in range(n_fam)]
        fam = rs.integers(0, n_fam, copies_total)
        pos = rs.integers(0, max(1, total_len - fam_len), copies_total)
        div = 0.05 + 0.10 * rs.uniform(copies_total)
        # vectorised insertion: build all copies, then scatter
        base = np.stack(cons)[fam]                                  # [copies][300]
        hit = rs.uniform(copies_total * fam_len).reshape(copies_total, fam_len) < div[:, None]
        shift = rs.integers(1, 4, copies_total * fam_len).reshape(copies_total, fam_len).astype(np.uint8)
        base = np.where(hit, (base + shift) & 3, base).astype(np.uint8)
        idx = pos[:, None] + np.arange(fam_len)[None, :]
        g[idx.reshape(-1)] = base.reshape(-1)
        # tandem arrays
        mono = random_bases(rs, 171, gc)
        n_arrays = max(1, int(0.02 * total_len / (171 * 1000)))
        for a in range(n_arrays):
            reps = 1000 if total_len > 400000 else max(2, total_len // (171 * 50))
            p = int(rs.integers(0, max(1, total_len - 171 * reps), 1)[0])
            arr = _mutate(rs, np.tile(mono, reps), 0.02)
            g[p:p + len(arr)] = arr
        # N runs
        long_run = int(min(1_000_000, total_len // 40))
        p = int(rs.integers(0, max(1, total_len - long_run), 1)[0])
        g[p:p + long_run] = 4
        for _ in range(5):
            p = int(rs.integers(0, max(1, total_len - 1000), 1)[0])
            g[p:p + 1000] = 4
    # cut into sequences of near-equal length
    cuts = [total_len * i // n_seqs for i in range(n_seqs + 1)]
    seqs = [g[cuts[i]:cuts[i + 1]] for i in range(n_seqs)]
    names = ["chr%d" % (i + 1) for i in range(n_seqs)]
    return seqs, names


def make_reads(seqs, n_reads, read_len, max_mm, seed, n_frac=0.001):
    """Reads sampled uniformly from the forward sequences, random strand, a number of substitutions
    uniform in {0..max_mm} at distinct positions, and one N in a fraction n_frac of the reads.
    Windows that contain a text N are re-drawn (an all-N read makes any FM-index search explode,
    the reference included).  Returns (codes [n_reads][read_len] uint8, offsets int64[n_reads+1]).
    """
    st = Stream(seed, 2)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    ok_len = np.maximum(lens - read_len + 1, 0)
    cum = np.concatenate(([0], np.cumsum(ok_len)))
    total = int(cum[-1])
    assert total > 0, "sequences shorter than the reads"
    genome = np.concatenate(seqs) if len(seqs) > 1 else seqs[0]
    seq_off = np.concatenate(([0], np.cumsum(lens)))
    ncum = np.concatenate(([0], np.cumsum(genome == 4, dtype=np.int64)))
    starts = np.empty(n_reads, dtype=np.int64)
    todo = np.arange(n_reads)
    while len(todo):
        u = st.integers(0, total, len(todo))
        sid = np.searchsorted(cum, u, side="right") - 1
        p = seq_off[sid] + (u - cum[sid])
        good = (ncum[p + read_len] - ncum[p]) == 0
        starts[todo[good]] = p[good]
        todo = todo[~good]
    reads = genome[starts[:, None] + np.arange(read_len)[None, :]].astype(np.uint8)
    # substitutions at distinct positions: rank random keys per read
    if max_mm > 0:
        nmm = st.integers(0, max_mm + 1, n_reads)
        keys = st.u64(n_reads * read_len).reshape(n_reads, read_len)
        order = np.argsort(keys, axis=1)[:, :max_mm]                     # max_mm distinct positions
        shift = st.integers(1, 4, n_reads * max_mm).reshape(n_reads, max_mm).astype(np.uint8)
        for k in range(max_mm):
            rows = np.nonzero(nmm > k)[0]
            cols = order[rows, k]
            reads[rows, cols] = (reads[rows, cols] + shift[rows, k]) & 3
    # one N in a few reads
    npick = st.uniform(n_reads) < n_frac
    ncol = st.integers(0, read_len, n_reads)
    rows = np.nonzero(npick)[0]
    reads[rows, ncol[rows]] = 4
    # reverse-complement half of the reads
    rc = st.uniform(n_reads) < 0.5
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    reads[rc] = comp[reads[rc][:, ::-1]]
    offsets = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return np.ascontiguousarray(reads), offsets


def codes_to_text(codes):
    return bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[np.asarray(codes, dtype=np.uint8)])


def write_fasta(path, seqs, names, width=60):
    with open(path, "wb") as f:
        for s, n in zip(seqs, names):
            f.write(b">" + n.encode() + b"\n")
            t = codes_to_text(s)
            for i in range(0, len(t), width):
                f.write(t[i:i + width] + b"\n")
