#!/usr/bin/env python3
"""Golden annotation tables from the UNMODIFIED reference (oracle/_ref/ref_driver ann = ann_build + ann_file_write,
src/index_ann.c:56-257, :384-434).  Run in the container that mounts /root/reference; only the outputs are committed:

  tests/golden/<text>.ann.k<K>.tau<T>.t<threads>.bin    the .ann file as the reference writes it (32-byte header + table)

Texts: repeats.fa (the text of test/test_index_ann.c:107-123), dense.fa and small.fa (with N), annfree.fa (made here:
two random sequences without N, with planted near-repeats, so that the reference's reuse of searches is exact)."""
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
         ("dense", 12, 1, 1), ("dense", 12, 2, 1), ("dense", 15, 3, 1), ("small", 24, 1, 1), ("small", 16, 2, 1)]


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


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    fa = os.path.join(OUT, "annfree.fa")
    if not os.path.exists(fa):
        make_annfree(fa)
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
