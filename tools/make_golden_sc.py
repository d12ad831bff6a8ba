#!/usr/bin/env python3
"""Golden vectors of blocksc_trail / scsearch_fw (src/blocksearch.c:7-103, :314-378) from the UNMODIFIED reference
(oracle/_ref/ref_driver sc / sctrail).  Run in the container that mounts /root/reference, after tools/make_golden.py
(it uses tests/golden/small.fa and repeats.fa); only the outputs (small data files) are committed:

  repeats.sc.reads.txt / repeats.sc.tau{0,1,2}.bin     the queries of test/test_blocksearch.c:33-36, :480-485 and more
  small.sc.k{24,31}.txt                                k-mers of small.fa in suffix-array order (a run of consecutive
                                                       distinct k-mers, as ann_build_mt visits them, then a sample of
                                                       the rest), k-mers with N, k-mers that are not in the text
  small.sc.k{K}.tau{T}.bin                             ref_driver sc      (trail = 0 for every query)
  small.sctrail.k{K}.tau{T}.bin                        ref_driver sctrail (a tree per strand class, trail = shared prefix)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapper_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
OUT = os.path.join(ROOT, "tests", "golden")
LETTERS = "ACGTN"


def read_fasta(path):
    seqs, cur = [], []
    for line in open(path):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip().upper())
    seqs.append("".join(cur))
    return seqs


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp()
    # ---- the reference's own example and the queries of its tests
    rbase = os.path.join(tmp, "rep")
    subprocess.check_call([REF, "build", os.path.join(OUT, "repeats.fa"), rbase])
    one = "ATCGATATCAGCCACTACGAGACAA"
    q = [one, "ATCGATATCAGCCACTACGAtACAA", "ATCGATATCAGCCACTACGAGACAc", "ATCGATATaAGCCACTACGAGACAA",
         "ATCGATATtAGCCACTACGAGACAA", "NNNNATATCAGCCACTACGAGACAA", "ATCGATATCAGgCACTACGAGACAA", "cTCGATATCAGCCACTACGAGACAA",
         "ATCGATATCAcCCACTACGAGACAA", revcomp(one), revcomp("ATCGATATCAGCCACTACGAGACAC"), "ATCGATATCAGCCACTACGAGACAN",
         "NTCGATATCAGCCACTACGAGACAA", "ATCGATATCAGCNACTACGAGACAA", one[:12], one[:13], one[5:], "A", "AT", "GGGGGGGGGGGGGGGGGGGGGGGGG"]
    rf = os.path.join(OUT, "repeats.sc.reads.txt")
    open(rf, "w").write("\n".join(q) + "\n")
    for tau in range(3):
        subprocess.check_call([REF, "sc", rbase, rf, str(tau), os.path.join(OUT, "repeats.sc.tau%d.bin" % tau)],
                              stderr=subprocess.DEVNULL)
    # ---- k-mers of the small synthetic reference in suffix-array order
    base = os.path.join(tmp, "small")
    subprocess.check_call([REF, "build", os.path.join(OUT, "small.fa"), base])
    seqs = read_fasta(os.path.join(OUT, "small.fa"))
    fwd = "$".join(seqs) + "$"
    text = fwd + revcomp(fwd[:-1].replace("$", "#")).replace("#", "$") + "$"     # F$rc(F)$, src/index_txt.c:314-334
    order = {c: i for i, c in enumerate("ACGTN")}
    rng = np.random.default_rng(5)
    for k in (24, 31):
        kmers = sorted({text[i:i + k] for i in range(len(text) - k + 1) if "$" not in text[i:i + k]},
                       key=lambda s: [order[c] for c in s])
        start = len(kmers) // 3
        pick = kmers[start:start + 500]                                       # consecutive: long trails
        pick += [kmers[i] for i in sorted(rng.choice(len(kmers), 400, replace=False))]   # scattered, still in order
        withn = [s for s in kmers if "N" in s]
        pick += withn[:40]
        for _ in range(30):                                                    # not in the text, and with N
            s = "".join(LETTERS[c] for c in rng.integers(0, 4, k))
            pick.append(s)
            t = list(s); t[int(rng.integers(0, k))] = "N"
            pick.append("".join(t))
        kf = os.path.join(OUT, "small.sc.k%d.txt" % k)
        open(kf, "w").write("\n".join(pick) + "\n")
        for tau in (1, 2, 3):
            subprocess.check_call([REF, "sc", base, kf, str(tau), os.path.join(OUT, "small.sc.k%d.tau%d.bin" % (k, tau))],
                                  stderr=subprocess.DEVNULL)
            subprocess.check_call([REF, "sctrail", base, kf, str(tau), os.path.join(OUT, "small.sctrail.k%d.tau%d.bin" % (k, tau))],
                                  stderr=subprocess.DEVNULL)
    # ---- a low-complexity text: k-mers with many neighbours (every branch of scsearch_fw is taken often)
    rng = np.random.default_rng(8)
    dseqs = []
    for _ in range(2):
        a = rng.choice(5, 1500, p=[0.46, 0.40, 0.07, 0.06, 0.01]).astype(np.uint8)
        a[700:760] = a[100:160]                                                # an exact repeat
        dseqs.append(a)
    dfa = os.path.join(OUT, "dense.fa")
    synth.write_fasta(dfa, dseqs, ["d1", "d2"])
    dbase = os.path.join(tmp, "dense")
    subprocess.check_call([REF, "build", dfa, dbase])
    seqs = read_fasta(dfa)
    fwd = "$".join(seqs) + "$"
    text = fwd + revcomp(fwd[:-1].replace("$", "#")).replace("#", "$") + "$"
    for k in (12, 15):
        kmers = sorted({text[i:i + k] for i in range(len(text) - k + 1) if "$" not in text[i:i + k]},
                       key=lambda s: [order[c] for c in s])
        start = len(kmers) // 4
        pick = kmers[start:start + 150] + [kmers[i] for i in sorted(rng.choice(len(kmers), 150, replace=False))]
        kf = os.path.join(OUT, "dense.sc.k%d.txt" % k)
        open(kf, "w").write("\n".join(pick) + "\n")
        for tau in (1, 2, 3):
            subprocess.check_call([REF, "sc", dbase, kf, str(tau), os.path.join(OUT, "dense.sc.k%d.tau%d.bin" % (k, tau))],
                                  stderr=subprocess.DEVNULL)
    print("blocksc golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
