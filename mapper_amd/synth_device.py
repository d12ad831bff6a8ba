"""Synthetic references and reads generated in HBM (torch on the GPU), for the configurations that are
too large for the numpy generator of synth.py (GRCh38 size: 3.1 Gbp, 100 M reads).

Same models as synth.py: i.i.d. background + interspersed repeat families +
tandem arrays + N runs, with the copy numbers of the chr21-size configuration kept as the text
grows: about 5 000 copies per family (so the number of families grows with the length) and one
monomer per tandem array (1 000 copies each).  One monomer shared by every array, as in synth.py,
would give a GRCh38-size text a 360 000-copy satellite and every read inside it 80 000 hits.
Reads are drawn uniformly from the forward sequences, random strand, a number of substitutions
uniform in {0..max_mm} at distinct positions, one N in a fraction of the reads.  The
streams are torch's Philox generator seeded per call, so a (seed, torch build) pair always gives the
same data; they are not the splitmix64 streams of synth.py.  Data generation only: torch is not on
the mapping path.
"""
import numpy as np
import torch

_CHUNK = 1 << 28


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def _random_bases(n, gc, gen, device):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    at = (1.0 - gc) / 2.0
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        u = torch.rand(hi - lo, generator=gen, device=device)
        out[lo:hi] = ((u >= at).to(torch.uint8) + (u >= at + gc / 2.0).to(torch.uint8) + (u >= at + gc).to(torch.uint8))
    return out


def make_genome(total_len, seed, n_seqs=1, gc=0.41, repeat_rich=True, device="cuda"):
    """Returns (codes: uint8 cuda tensor of total_len symbols, lens: int64 numpy [n_seqs], names)."""
    gen = _gen(seed, device)
    g = _random_bases(total_len, gc, gen, device)
    if repeat_rich:
        fam_len = 300
        copies_total = max(20, int(0.10 * total_len / fam_len))
        n_fam = max(20, copies_total // 5000)
        cons = _random_bases(n_fam * fam_len, gc, gen, device).view(n_fam, fam_len)
        ar = torch.arange(fam_len, device=device)
        step = 1 << 20
        for lo in range(0, copies_total, step):
            k = min(step, copies_total - lo)
            fam = torch.randint(0, n_fam, (k,), generator=gen, device=device)
            pos = torch.randint(0, max(1, total_len - fam_len), (k,), generator=gen, device=device)
            div = 0.05 + 0.10 * torch.rand(k, generator=gen, device=device)
            base = cons[fam]
            hit = torch.rand(k, fam_len, generator=gen, device=device) < div[:, None]
            shift = torch.randint(1, 4, (k, fam_len), generator=gen, device=device, dtype=torch.uint8)
            base = torch.where(hit, (base + shift) & 3, base)
            # copies may overlap: the deterministic index_put keeps the data a function of the seed alone
            prev = torch.are_deterministic_algorithms_enabled()
            torch.use_deterministic_algorithms(True)
            try:
                g.index_put_(((pos[:, None] + ar[None, :]).reshape(-1),), base.reshape(-1))
            finally:
                torch.use_deterministic_algorithms(prev)
        n_arrays = max(1, int(0.02 * total_len / (171 * 1000)))
        monos = _random_bases(171 * n_arrays, gc, gen, device).view(n_arrays, 171)
        reps = 1000 if total_len > 400000 else max(2, total_len // (171 * 50))
        starts = torch.randint(0, max(1, total_len - 171 * reps), (n_arrays,), generator=gen, device=device).tolist()
        for a, p in enumerate(starts):
            arr = monos[a].repeat(reps)
            hit = torch.rand(arr.numel(), generator=gen, device=device) < 0.02
            shift = torch.randint(1, 4, (arr.numel(),), generator=gen, device=device, dtype=torch.uint8)
            g[p:p + arr.numel()] = torch.where(hit, (arr + shift) & 3, arr)
        long_run = int(min(1_000_000, total_len // 40))
        p = int(torch.randint(0, max(1, total_len - long_run), (1,), generator=gen, device=device).item())
        g[p:p + long_run] = 4
        for p in torch.randint(0, max(1, total_len - 1000), (5,), generator=gen, device=device).tolist():
            g[p:p + 1000] = 4
    cuts = [total_len * i // n_seqs for i in range(n_seqs + 1)]
    lens = np.diff(np.array(cuts, dtype=np.int64))
    names = ["chr%d" % (i + 1) for i in range(n_seqs)]
    return g, lens, names


def make_reads(genome, lens, n_reads, read_len, max_mm, seed, n_frac=0.001, return_origin=False):
    """genome: uint8 cuda tensor (the sequences one after another), lens: their lengths.
    Returns codes [n_reads][read_len] uint8 (cuda tensor); with return_origin also (start, strand):
    the window's first position in the concatenated sequences and 1 where the read is its reverse
    complement."""
    device = genome.device
    gen = _gen(seed, device)
    total = genome.numel()
    bounds = torch.as_tensor(np.cumsum(lens), device=device)           # exclusive ends of the sequences
    ar = torch.arange(read_len, device=device)
    reads = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    todo = torch.arange(n_reads, device=device)
    origin = torch.empty(n_reads, dtype=torch.int64, device=device)
    strand = torch.zeros(n_reads, dtype=torch.uint8, device=device)
    step = 1 << 22
    while todo.numel():
        nxt = []
        for lo in range(0, todo.numel(), step):
            t = todo[lo:lo + step]
            p = torch.randint(0, total - read_len + 1, (t.numel(),), generator=gen, device=device)
            w = genome[p[:, None] + ar[None, :]]
            same_seq = torch.searchsorted(bounds, p, right=True) == torch.searchsorted(bounds, p + read_len - 1, right=True)
            good = same_seq & ~(w == 4).any(dim=1)          # windows with a text N are re-drawn (synth.py)
            reads[t[good]] = w[good]
            origin[t[good]] = p[good]
            nxt.append(t[~good])
        todo = torch.cat(nxt)
    for lo in range(0, n_reads, step):
        k = min(step, n_reads - lo)
        r = reads[lo:lo + k]
        if max_mm > 0:
            nmm = torch.randint(0, max_mm + 1, (k,), generator=gen, device=device)
            cols = torch.randint(0, read_len, (k, max_mm), generator=gen, device=device)
            while True:                                      # distinct positions per read
                s, _ = torch.sort(cols, dim=1)
                dup = (s[:, 1:] == s[:, :-1]).any(dim=1) if max_mm > 1 else torch.zeros(k, dtype=torch.bool, device=device)
                nd = int(dup.sum().item())
                if nd == 0:
                    break
                cols[dup] = torch.randint(0, read_len, (nd, max_mm), generator=gen, device=device)
            shift = torch.randint(1, 4, (k, max_mm), generator=gen, device=device, dtype=torch.uint8)
            rows = torch.arange(k, device=device)
            for j in range(max_mm):
                sel = rows[nmm > j]
                c = cols[sel, j]
                r[sel, c] = (r[sel, c] + shift[sel, j]) & 3
        npick = torch.rand(k, generator=gen, device=device) < n_frac
        ncol = torch.randint(0, read_len, (k,), generator=gen, device=device)
        sel = torch.nonzero(npick).flatten()
        r[sel, ncol[sel]] = 4
        rc = torch.rand(k, generator=gen, device=device) < 0.5
        comp = torch.tensor([3, 2, 1, 0, 4], dtype=torch.uint8, device=device)
        sel = torch.nonzero(rc).flatten()
        r[sel] = comp[r[sel].flip(1).long()]
        strand[lo + sel] = 1
        reads[lo:lo + k] = r
    return (reads, origin, strand) if return_origin else reads
