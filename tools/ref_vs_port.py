#!/usr/bin/env python3
"""Conversion factor of the CPU baseline: the UNMODIFIED reference (oracle/_ref/ref_driver,
compiled in place from /root/reference/src) against the oracle's C restatement, one thread each, same index
files, same reads, outputs compared.  Runs only where /root/reference is mounted; prints JSON lines."""
import json, os, subprocess, sys, tempfile, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapper_amd import synth                 # noqa: E402
from oracle import pyoracle as po            # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
CONFIGS = [("config 1: 1 Mbp random, 10 k x 50 nt, tau 0", 1_000_000, 0.5, 1, 10_000, 50, 0),
           ("config 2 (sample): 4.64 Mbp random, 20 k x 100 nt, tau 2", 4_641_652, 0.508, 2, 20_000, 100, 2)]


def main():
    tmp = tempfile.mkdtemp(prefix="ref_vs_port_")
    for desc, L, gc, seed, n, m, tau in CONFIGS:
        seqs, names = synth.make_genome(L, seed, n_seqs=1, gc=gc, repeat_rich=False)
        fa = os.path.join(tmp, "g.fa")
        synth.write_fasta(fa, seqs, names)
        base = os.path.join(tmp, "g%d" % seed)
        subprocess.check_call([REF, "build", fa, base], stderr=subprocess.DEVNULL)
        codes, offs = synth.make_reads(seqs, n, m, tau, seed=seed + 100)
        rfile = os.path.join(tmp, "reads.txt")
        with open(rfile, "w") as f:
            for c in codes:
                f.write(synth.codes_to_text(c).decode() + "\n")
        out = subprocess.check_output([REF, "map", base, rfile, str(tau), os.path.join(tmp, "out.bin"), "1"],
                                      stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
        ref = json.loads(out)
        d = po.Index(base).map_batch(codes.reshape(-1), offs, tau, threads=1)
        dump = po.HitDump.load(os.path.join(tmp, "out.bin"))
        same = bool(np.array_equal(d.located(), dump.located()))
        print(json.dumps({"config": desc, "reference_reads_per_s": ref["reads_per_s"], "port_reads_per_s": n / d.seconds,
                          "port_over_reference": (n / d.seconds) / ref["reads_per_s"], "threads": 1,
                          "located_identical": same}), flush=True)


if __name__ == "__main__":
    main()
