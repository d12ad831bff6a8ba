#!/usr/bin/env python3
"""Generates tests/golden/* with the UNMODIFIED reference (oracle/_ref/ref_driver, compiled in place
from /root/reference/src by `make -C oracle ref`).  Run in the container that mounts /root/reference;
only the outputs (small data files) are committed:

  small.fa            synthetic two-sequence reference with repeats, an N run and a short tandem array
  small.reads.txt     reads: planted with 0..4 substitutions on both strands, ragged lengths, N's,
                      and edge cases (too short for tau+1 blocks, 256 and 257 letters, unmappable)
  small.tau{0..4}.bin hit dumps of the reference (format: oracle/ref_driver.c header)
  small.sha256.json   sha256 of the four index files the reference writes for small.fa
  repeats.fa          = test/examples/repeats.fa of the reference (a data file its tests use)
  repeats.tau{0,1,2}.bin
  kat_*.txt           answers of the reference primitives on the literal texts of its unit tests
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapper_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp()

    # ---- small synthetic reference
    seqs, names = synth.make_genome(24000, seed=11, n_seqs=2, gc=0.45, repeat_rich=True)
    seqs = [s.copy() for s in seqs]
    seqs[0][100:160] = seqs[1][500:560]            # an exact 60-mer shared by both sequences
    seqs[1][3000:3004] = 4                         # a short N run inside mappable sequence
    fa = os.path.join(OUT, "small.fa")
    synth.write_fasta(fa, seqs, ["seqA", "seqB some description"])
    base = os.path.join(tmp, "small")
    subprocess.check_call([REF, "build", fa, base])
    sha = {ext: hashlib.sha256(open(base + "." + ext, "rb").read()).hexdigest() for ext in ("sym", "txt", "sar", "bwt")}
    json.dump(sha, open(os.path.join(OUT, "small.sha256.json"), "w"), indent=1)

    reads = []
    rng = synth.Stream(12, 0)
    for m, cnt in ((100, 120), (50, 60), (36, 30), (150, 40), (75, 20), (256, 4), (25, 10)):
        codes, _ = synth.make_reads(seqs, cnt, m, 4, seed=100 + m, n_frac=0.1)
        reads += [synth.codes_to_text(c).decode() for c in codes]
    genome_txt = synth.codes_to_text(np.concatenate(seqs)).decode()
    reads += [genome_txt[2990:3090], genome_txt[2998:3030]]          # reads crossing the short N run
    reads += ["ACGT", "AC", "ACGTA", "NACGTACGTAN", "ACGTNNACGTAC", "NNNNACGTACGTNNNN", "acgtnACGTN", "",
              "ACGTRYKMACGTXXACGT", genome_txt[100:160], genome_txt[100:160].lower(),
              "T" * 40, "A" * 257, genome_txt[5000:5256], genome_txt[5000:5257]]
    # a few random (unmappable) reads
    reads += [synth.codes_to_text(synth.random_bases(rng, 80)).decode() for _ in range(5)]
    rfile = os.path.join(OUT, "small.reads.txt")
    open(rfile, "w").write("\n".join(reads) + "\n")
    for tau in range(5):
        subprocess.check_call([REF, "map", base, rfile, str(tau), os.path.join(OUT, "small.tau%d.bin" % tau)],
                              stderr=subprocess.DEVNULL)

    # ---- the reference's own example
    rep = os.path.join(OUT, "repeats.fa")
    shutil.copyfile("/root/reference/test/examples/repeats.fa", rep)
    rbase = os.path.join(tmp, "rep")
    subprocess.check_call([REF, "build", rep, rbase])
    rreads = ["ATCGATATCAGCCACTACGAGACAA", "ATCGATATCAGCCACTACGAtACAA", "ATCGATATCAGCCACTACGAGACAc",
              "ATCGATATaAGCCACTACGAGACAA", "ATCGATATtAGCCACTACGAGACAA", "NNNNATATCAGCCACTACGAGACAA",
              "TTGTCTCGTAGTGGCTGATATCGAT"]
    rrf = os.path.join(OUT, "repeats.reads.txt")
    open(rrf, "w").write("\n".join(rreads) + "\n")
    for tau in range(3):
        subprocess.check_call([REF, "map", rbase, rrf, str(tau), os.path.join(OUT, "repeats.tau%d.bin" % tau)],
                              stderr=subprocess.DEVNULL)

    # ---- literal texts of the reference's unit tests, answered by the reference primitives
    kats = {"bwt32": "TTAGCAGTAGTCGTA$TACGACTACTGCTAA$",
            "periodic": "ACGTTGCA$" * 1000 + "TGCAACGT$" * 1000}
    for name, spec in kats.items():
        subprocess.check_call([REF, "kat", spec, os.path.join(OUT, "kat_%s.txt" % name)])
    shutil.rmtree(tmp)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
