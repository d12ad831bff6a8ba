#!/usr/bin/env python3
"""Golden annotation tables from the UNMODIFIED reference (oracle/_ref/ref_driver ann = ann_build + ann_file_write,
src/index_ann.c:56-257, :384-434).  Run in the container that mounts /root/reference; only the outputs are committed:

  tests/golden/<text>.ann.k<K>.tau<T>.t<threads>.bin    the .ann file as the reference writes it (32-byte header + table)

Texts: repeats.fa (the text of test/test_index_ann.c:107-123), dense.fa and small.fa (with N), annfree.fa (made here:
two random sequences without N, with planted near-repeats, so that the reference's reuse of searches is exact), annN.fa
(made here: N in every few k-mers, where that reuse is stale)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
OUT = os.path.join(ROOT, "tests", "golden")
CASES = [("repeats", 25, 1, 1), ("repeats", 25, 2, 1), ("repeats", 10, 1, 1), ("repeats", 10, 1, 2), ("repeats", 12, 3, 1),
         ("annfree", 20, 1, 1), ("annfree", 14, 2, 1), ("annfree", 12, 3, 1), ("annfree", 14, 2, 2),
         ("dense", 12, 1, 1), ("dense", 12, 2, 1), ("dense", 15, 3, 1), ("small", 24, 1, 1), ("small", 16, 2, 1),
         ("annN", 8, 3, 1), ("annN", 9, 2, 1), ("annN", 12, 3, 1), ("annN", 13, 2, 3), ("annN", 16, 1, 1)]


def make_annfree(path):
    rng = np.random.default_rng(77)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for L in (5000, 3000):
        s = letters[rng.integers(0, 4, L)].copy()
        for _ in range(L // 150):                       # near-repeats: a copy of a piece with a few changed letters
            a, b, m = rng.integers(0, L - 60), rng.integers(0, L - 60), int(rng.integers(20, 60))
            s[b:b + m] = s[a:a + m]
            for _ in range(int(rng.integers(0, 4))):
                s[b + rng.integers(0, m)] = letters[rng.integers(0, 4)]
        seqs.append(s.tobytes().decode())
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">free%d\n" % i)
            for k in range(0, len(s), 70):
                f.write(s[k:k + 70] + "\n")


def make_ann_n(path):
    """Three sequences with N in every few k-mers (2 %, 5 %, 0.4 % of the letters) and planted near-repeats: next to k-mers
    with N the reference's reuse of the last k-mer's search is stale (src/blocksearch.c:41-43, :63-66, :73-86), so these
    tables pin that behaviour (a fresh search per k-mer differs in up to a fifth of the bytes)."""
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "w") as f:
        for i, (L, p_n) in enumerate(((9000, 0.02), (7000, 0.05), (4000, 0.004))):
            s = rng.integers(0, 4, L).astype(np.uint8)
            for _ in range(L // 120):
                a, b, m = rng.integers(0, L - 50), rng.integers(0, L - 50), int(rng.integers(15, 50))
                s[b:b + m] = s[a:a + m]
                for _ in range(int(rng.integers(0, 3))):
                    s[b + rng.integers(0, m)] = rng.integers(0, 4)
            s[rng.random(L) < p_n] = 4
            t = letters[s].tobytes().decode()
            f.write(">n%d\n" % i)
            for k in range(0, len(t), 70):
                f.write(t[k:k + 70] + "\n")


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    fa = os.path.join(OUT, "annfree.fa")
    if not os.path.exists(fa):
        make_annfree(fa)
    fa = os.path.join(OUT, "annN.fa")
    if not os.path.exists(fa):
        make_ann_n(fa)
    tmp = tempfile.mkdtemp()
    built = {}
    for name, k, tau, th in CASES:
        if name not in built:
            built[name] = os.path.join(tmp, name)
            subprocess.check_call([REF, "build", os.path.join(OUT, name + ".fa"), built[name]], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = os.path.join(OUT, "%s.ann.k%d.tau%d.t%d.bin" % (name, k, tau, th))
        subprocess.check_call([REF, "ann", built[name], str(k), str(tau), str(th), out], stderr=subprocess.DEVNULL)
        print(out, os.path.getsize(out))


if __name__ == "__main__":
    sys.exit(main())
