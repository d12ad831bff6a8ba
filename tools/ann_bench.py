#!/usr/bin/env python3
"""Rate of gpumap_ann_build (ann_build, src/index_ann.c:56-257: one blocksc_trail search per distinct k-mer of the text
plus the folding of the hits into the annotation table) against the CPU: the unmodified reference's own ann_build
(oracle/_ref/ref_driver ann, its pthreads) where that binary travelled with the tree, and the oracle's restatement
(one thread).  Same index files for all three; the tables are compared.  Prints one JSON line.

    python tools/ann_bench.py [--genome-len 4000000] [--kmer 25] [--tau 1] [--threads 16] [--random]
"""
import argparse, json, os, subprocess, sys, tempfile, time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapper_amd as ma
from mapper_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=4_000_000)
    ap.add_argument("--kmer", type=int, default=25)
    ap.add_argument("--tau", type=int, default=1)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--random", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--device-build", action="store_true", help="generate and index the text in HBM (large texts; no CPU side then)")
    ap.add_argument("--jump-gb", type=int, default=0, help="cap of the jump table in GiB (0 = the library's default)")
    ap.add_argument("--oracle-max", type=int, default=3_000_000, help="largest text the one-thread restatement is timed on")
    args = ap.parse_args()
    work = tempfile.mkdtemp(prefix="gpumap_ann_")
    base = os.path.join(work, "ref")
    if args.device_build:
        import torch
        from mapper_amd import synth_device
        g, lens, names = synth_device.make_genome(args.genome_len, 5, n_seqs=24, gc=0.41, repeat_rich=not args.random)
        ctx = ma.Context.build_device(lens, 0, keep_reference=False, device_ptr=g.data_ptr())
        del g
        torch.cuda.empty_cache()
        args.no_cpu = True
        n_seqs = 24
    else:
        seqs, names = synth.make_genome(args.genome_len, 5, n_seqs=2, gc=0.41, repeat_rich=not args.random)
        index = ma.HostIndex.build_codes(seqs, names, base)
        ctx = ma.Context.from_index(index, 0)
        n_seqs = 2
        ctx.ann_build(args.kmer, args.tau)                   # warm-up (allocations)
    if args.jump_gb:
        ctx.set_option("jump_bytes", args.jump_gb << 30)
    t = time.perf_counter()
    table, st = ctx.ann_build(args.kmer, args.tau)
    gpu_s = time.perf_counter() - t
    out = {"metric": "gpumap_ann_build: k-mers annotated/s", "kmer": args.kmer, "tau": args.tau, "text_len": int(2 * (args.genome_len + n_seqs)),
           "genome": "random" if args.random else "repeat-rich", "kmers_searched": st["queries"], "ext_steps": st["n_ext"],
           "hits": st["hits"], "annotated_positions": int(np.count_nonzero(table)),
           "gpu": {"seconds": gpu_s, "device_seconds": st["us"] * 1e-6, "kmers_per_s": st["queries"] / gpu_s}}
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_driver")
    if not args.no_cpu and os.access(ref, os.X_OK):
        path = os.path.join(work, "ref.ann")
        t = time.perf_counter()
        subprocess.check_call([ref, "ann", base, str(args.kmer), str(args.tau), str(args.threads), path], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        cpu_s = time.perf_counter() - t
        rt = np.fromfile(path, dtype=np.uint8)[32:]
        out["cpu_reference"] = {"kind": "reference", "threads": args.threads, "seconds": cpu_s, "kmers_per_s": st["queries"] / cpu_s,
                                "table_bytes_that_differ": int(np.count_nonzero(rt != table)),
                                "note": "ann_build of the unmodified reference (index load included); its table depends on `threads` "
                                        "(job prefixes) and on its reuse of searches next to k-mers with N"}
        out["gpu"]["speedup_over_reference"] = cpu_s / gpu_s
    if not args.no_cpu and out["text_len"] <= args.oracle_max:
        from oracle import pyoracle as po
        orc = po.Index(base)
        t = time.perf_counter()
        want, ost = orc.ann_build(args.kmer, args.tau, 1, trail_reuse=False)
        out["cpu_port"] = {"kind": "port", "threads": 1, "seconds": time.perf_counter() - t, "identical": bool(np.array_equal(want, table)),
                           "ext_steps_identical": ost["n_ext"] == st["n_ext"]}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
