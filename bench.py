#!/usr/bin/env python3
"""bench.py -- throughput of the query hot path (block search + locate) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One process per GPU (for N > 1 launched by torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* come from the environment).  Rank 0 builds the synthetic index on the host, uploads and
repacks it on its GPU and broadcasts the three device arrays once over xGMI with RCCL
(torch.distributed, backend "nccl"); every rank then maps its own shard of reads against its own
replica.  No collective on the data path: reads are independent units (weak scaling: the per-GPU
batch is fixed, value = reads of all ranks / max-over-ranks time).

A "step" is one pass of the hot path over one chunk of reads whose letters are resident in HBM when the
timed region starts: encode (k_read_lengths, k_encode) + search (k_search, k_verify) +
suffix-array locate + ordering + the copy of the hits to page-locked host memory (packed results: per hit, in
read order and by ascending position, 5 bytes = text position and score; per read a 4-byte offset and a status
byte; --results compact: 8-byte keys, 8-byte offsets and status).  Steps cycle over 8 distinct chunks and go through the library's pipeline
(gpumap_pipe), so that the copy of one chunk's hits overlaps the next chunk's kernels.
Default workload: the GRCh38-size index of config 4 (synthetic, generated and indexed in HBM).

One JSON line is printed by rank 0; see DESIGN.md section "Measurement" for the accounting of
`roofline` (algorithmic bytes = 128 B per extension step of the reference schedule, i.e. two
64-byte Occ touches) and `cpu_baseline` (the oracle's C restatement on the host
cores, bounded sample).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: genome length, sequences, repeat-rich, seed, reads per GPU, read length, tau
    "plumbing": dict(L=1_000_000, n_seqs=1, rich=False, gc=0.5, seed=1, reads=10_000, m=50, tau=0,
                     desc="config 1: 1 Mbp random, 10k x 50 nt, exact"),
    "ecoli": dict(L=4_641_652, n_seqs=1, rich=False, gc=0.508, seed=2, reads=1_000_000, m=100, tau=2,
                  desc="config 2 stand-in: 4.64 Mbp random (E. coli K-12 size), 1M x 100 nt, <=2 mismatches"),
    "chr21": dict(L=46_709_983, n_seqs=1, rich=True, gc=0.41, seed=3, reads=10_000_000, m=100, tau=2,
                  desc="config 3 stand-in: 46.7 Mbp repeat-rich (chr21 size), 10M x 100 nt, <=2 mismatches"),
    # GRCh38 size: generated and indexed in HBM (mapper_amd/synth_device.py, gpumap_build_device); the reads of
    # config 4 (100 M) are the 8-GPU total, 12.5 M per GPU (weak scaling: the per-GPU batch is fixed)
    "grch38": dict(L=3_100_000_000, n_seqs=24, rich=True, gc=0.41, seed=4, reads=12_500_000, m=100, tau=2, device=True,
                   desc="config 4 stand-in: 3.1 Gbp repeat-rich in 24 sequences (GRCh38 size), 12.5M x 100 nt per GPU, <=2 mismatches"),
    "grch38_150": dict(L=3_100_000_000, n_seqs=24, rich=True, gc=0.41, seed=4, reads=4_000_000, m=150, tau=4, device=True,
                       desc="config 5 stand-in: GRCh38-size index, 4M x 150 nt per GPU, <=4 mismatches"),
    "g1": dict(L=1_000_000_000, n_seqs=8, rich=True, gc=0.41, seed=5, reads=12_500_000, m=100, tau=2, device=True,
               desc="1 Gbp repeat-rich in 8 sequences, 12.5M x 100 nt per GPU, <=2 mismatches"),
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
BYTES_PER_EXT = 128.0           # two 64-B Occ touches per extension step


class _DevPtr:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("GPUMAP_WORKLOAD", "grch38"))
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's)")
    ap.add_argument("--chunks", type=int, default=8, help="distinct chunks of reads per GPU resident in HBM")
    ap.add_argument("--depth", type=int, default=4, help="batch objects of the pipeline (a result comes back 2.8 steps after its chunk "
                    "went in: with 3 there is no slack and results come in bursts, --results compact loses 2 %% to that)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive legs (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--verify", type=int, default=-1, help="option 'verify' of the library (-1 = its default by text size)")
    ap.add_argument("--verify-tile", type=int, default=0, help="option 'verify_tile' of the library (0 = its default)")
    ap.add_argument("--jump-k", type=int, default=-2, help="option 'jump_k' of the library (-2 = leave its default)")
    ap.add_argument("--results", default="packed", choices=["packed", "compact"],
                    help="form of the results that cross the link in the timed region (GPUMAP_PIPE_PACKED / GPUMAP_PIPE_COMPACT)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="gpumap_set_option on the context (A/B runs)")
    ap.add_argument("--device-build", action="store_true", help="generate and index in HBM also for the small workloads")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 "
                    "on a box with fewer GPUs than ranks: ranks then share devices)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path with any N, also 1: process group (RCCL), broadcast of the index arrays, a replica "
                         "created from the broadcast buffers (gpumap_create_from_device) that serves the timed region, replica check")
    args = ap.parse_args()
    # stdout carries the one JSON line and nothing else: libraries that write there (RCCL prints its version banner to
    # stdout when a communicator is made) go to stderr for the length of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import mapper_amd as ma
    from mapper_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if args.backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    def note(msg):
        if os.environ.get("GPUMAP_BENCH_VERBOSE"):
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    w = dict(WORKLOADS[args.workload])
    if args.reads:
        w["reads"] = args.reads
    on_device = bool(w.get("device")) or args.device_build
    t_setup = time.time()
    tmp = tempfile.mkdtemp(prefix="gpumap_bench_")
    base = os.path.join(tmp, args.workload)
    keep = []
    t_index = t_upload = t_bcast = t_replica = 0.0
    want_cpu = not args.no_cpu_baseline and world == 1
    read_seed = w["seed"] * 1000 + 17 + rank
    if on_device:
        from mapper_amd import synth_device
        # every rank generates the same synthetic sequences in its own HBM to draw its reads from
        g, lens, names = synth_device.make_genome(w["L"], w["seed"], n_seqs=w["n_seqs"], gc=w["gc"], repeat_rich=w["rich"])
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        note("sequences generated")
        if rank == 0:
            t0 = time.time()
            ctx = ma.Context.build_device(lens, local, keep_reference=want_cpu or dist_on, device_ptr=g.data_ptr())
            t_index = time.time() - t0
        seqs = None
    else:
        # every rank regenerates the same synthetic sequences (deterministic) to draw its reads from
        seqs, names = synth.make_genome(w["L"], w["seed"], n_seqs=w["n_seqs"], gc=w["gc"], repeat_rich=w["rich"])
        if rank == 0:
            t0 = time.time()
            hix = ma.HostIndex.build_codes(seqs, names, base)       # also writes base.* for the CPU baseline
            t_index = time.time() - t0
            t0 = time.time()
            ctx = ma.Context.from_index(hix, local)
            t_upload = time.time() - t0
    root_ctx = ctx if rank == 0 else None
    if dist_on:
        # one-off replication of the device-resident index: RCCL broadcast over xGMI
        t0 = time.time()
        if rank == 0:
            dev, hdr = ctx.export()
            meta = [bytes(hdr), int(dev.occ_bytes), int(dev.sa_bytes), int(dev.tables_bytes), int(dev.text_bytes)]
        else:
            meta = [None, 0, 0, 0, 0]
        note("index ready, broadcasting")
        dist.broadcast_object_list(meta, src=0)
        bufs = []
        for i, nbytes in enumerate(meta[1:]):
            if nbytes == 0:
                bufs.append(None)
                continue
            if rank == 0:
                ptr = (dev.occ_blocks, dev.sa_words, dev.tables, dev.text)[i]
                t = torch.as_tensor(_DevPtr(ptr, nbytes), device=torch.device("cuda", local))
            else:
                t = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", local))
            piece = 1 << 30                                  # 1-GiB pieces: counts stay far below 2^31 elements
            for lo in range(0, nbytes, piece):
                dist.broadcast(t[lo:min(nbytes, lo + piece)], src=0)
            bufs.append(t)
            note("broadcast of buffer %d done (%d bytes)" % (i, nbytes))
        torch.cuda.synchronize()
        t_bcast = time.time() - t0
        if rank != 0 or args.force_dist:
            # (--force-dist on rank 0: the replica takes the root's place; the root keeps the arrays alive and gives
            # up its own jump table, so that two of them do not sit in one GPU's memory)
            t0 = time.time()
            if rank == 0:
                root_ctx.set_option("jump_k", 0)
            d = ma.DevIndex(bufs[0].data_ptr(), meta[1], bufs[1].data_ptr(), meta[2], bufs[2].data_ptr(), meta[3],
                            bufs[3].data_ptr() if bufs[3] is not None else None, meta[4])
            ctx = ma.Context.from_device(d, meta[0], local, keep=bufs)
            t_replica = time.time() - t0
        keep = bufs
    if args.waves_per_cu:
        ctx.set_option("waves_per_cu", args.waves_per_cu)
    if args.verify >= 0:
        ctx.set_option("verify", args.verify)
    if args.verify_tile > 0:
        ctx.set_option("verify_tile", args.verify_tile)
    if args.jump_k != -2:
        ctx.set_option("jump_k", args.jump_k)
    for o in args.opt:
        ctx.set_option(o.split("=")[0], int(o.split("=")[1]))

    # this rank's shard: its own reads (weak scaling).  n_chunks distinct chunks of reads_per_gpu reads each, their
    # LETTERS and offsets resident in HBM before the timed region starts (8 x 12.5 M = the 100 M reads of config 4
    # on one GPU); the first chunk also on the host (CPU baseline, parity sample, PCIe-inclusive figures)
    ascii_lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device="cuda")

    def to_letters(rd):
        flat = rd.reshape(-1)
        out = torch.empty_like(flat)
        for lo in range(0, flat.numel(), 1 << 27):
            out[lo:lo + (1 << 27)] = ascii_lut[flat[lo:lo + (1 << 27)].long()]
        return out

    if not on_device:
        from mapper_amd import synth_device
        g = torch.from_numpy(np.concatenate(seqs)).cuda()
        if not bool((g.cpu().numpy() == np.concatenate(seqs)).all()):
            sys.stderr.write("bench.py: the copy of the sequences in HBM differs from the host's\n")
            sys.exit(1)
        lens = np.array([len(x) for x in seqs], dtype=np.int64)
    d_offs = (torch.arange(w["reads"] + 1, dtype=torch.int64, device="cuda") * w["m"]).contiguous()
    d_chunks = []
    codes = None
    for k in range(args.chunks):
        rd = synth_device.make_reads(g, lens, w["reads"], w["m"], w["tau"], seed=read_seed + 7919 * k)
        if k == 0:
            codes = rd.cpu().numpy()
        d_chunks.append(to_letters(rd))
        del rd
    offs = np.arange(w["reads"] + 1, dtype=np.int64) * w["m"]
    # N > 1: one small sample that every rank maps against its own replica (the same reads on every rank)
    sample = synth_device.make_reads(g, lens, min(20000, w["reads"]), w["m"], w["tau"], seed=w["seed"] * 1000 + 5).cpu().numpy() \
        if dist_on else None
    del g
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t_setup = time.time() - t_setup
    note("reads resident, timing")

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: encode + search + locate + order + D2H of the hits, chunk after chunk through
    # the library's pipeline (gpumap_pipe: `depth` batch objects with streams of their own, so that the copy of one
    # chunk's results overlaps the next chunk's kernels); inputs = letters in HBM, outputs = packed results (5 bytes per
    # hit: text position and score, in read order; 4-byte offsets and a status byte per read) in page-locked host memory.
    lib = ma.lib()
    result_flag = ma.PIPE_PACKED if args.results == "packed" else ma.PIPE_COMPACT
    pipe = ctx.pipe(depth=args.depth, tau=w["tau"], flags=ma.SRC_TEXT | ma.SRC_DEVICE | result_flag)
    n_reads = w["reads"]

    done_at = []                                # when each chunk's result came back (the spread of the steps goes into the line)

    def run_steps(k_steps, first=0):
        located = 0
        taken = 0
        for k in range(k_steps):
            if k - taken == args.depth:
                raw, _ = pipe.next_raw()
                done_at.append(time.perf_counter())
                located += int(raw.n_locs)
                pipe.release_raw(raw)
                taken += 1
            pipe.submit(d_chunks[(first + k) % len(d_chunks)].data_ptr(), d_offs.data_ptr(), n_reads)
        while taken < k_steps:
            raw, _ = pipe.next_raw()
            done_at.append(time.perf_counter())
            located += int(raw.n_locs)
            pipe.release_raw(raw)
            taken += 1
        return located

    run_steps(args.depth)                       # every slot allocates its device buffers and its result arena (set-up)
    run_steps(args.warmup)
    barrier()
    del done_at[:]
    t0 = time.perf_counter()
    located_total = run_steps(args.steps, first=args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    # every read was drawn from the indexed sequences with at most tau substitutions, so every read is located at least
    # once (at its origin): a run that located fewer mapped something else than it says, and must not print a line
    if located_total < 0.9 * n_reads * args.steps:
        sys.stderr.write("bench.py: %d located hits for %d reads drawn from the text: the reads and the index do not belong together\n"
                         % (located_total, n_reads * args.steps))
        sys.exit(1)
    if dist_on:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- the same work one chunk at a time on one batch object, for the per-kernel times (in the pipeline the
    # kernels of different chunks overlap, so their event times are not per-kernel times)
    batch = ctx.new_batch()
    ctx.set_option("compact_results", 1)
    per = {k: [] for k in ("encode", "k_search", "k_verify", "locate", "order", "device_total")}
    for i in range(max(2, min(args.steps, 4)) + 1):
        batch.stage(d_chunks[i % len(d_chunks)].data_ptr(), (d_offs.data_ptr(), n_reads), ma.SRC_TEXT | ma.SRC_DEVICE)
        t1 = time.perf_counter()
        c = batch.run(w["tau"])
        dt = time.perf_counter() - t1
        if i == 0:
            continue                            # (allocations)
        e = batch.stats_ext()
        per["encode"].append(e["us_encode"] / 1000.0)
        per["k_search"].append(c["us_k_search"] / 1000.0)
        per["k_verify"].append(c["us_k_verify"] / 1000.0)
        per["locate"].append(c["ms_locate"])
        per["order"].append(c["ms_sort"])
        per["device_total"].append(1e3 * dt)
    e = batch.stats_ext()
    ctx.set_option("compact_results", 0)
    # replicas: every rank maps the same small sample; the checksums of the sorted keys must agree
    replica = None
    if dist_on:
        ns = len(sample)
        h = ctx.map_batch(sample.reshape(-1), offs[:ns + 1], w["tau"])
        mine = np.array([int(h.n_locs), int(np.bitwise_xor.reduce(h.located().reshape(-1) * np.int64(0x9E3779B97F4A7C15 >> 1))) if h.n_locs else 0],
                        dtype=np.int64)
        allv = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(world)]
        dist.all_gather(allv, torch.from_numpy(mine).cuda())
        allv = [tuple(v.cpu().tolist()) for v in allv]
        replica = {"sample_reads": ns, "located": int(mine[0]), "identical_across_ranks": len(set(allv)) == 1}
        if rank == 0 and not args.no_cpu_baseline:
            # and against the oracle (the checker: test infrastructure), on the root's reference-layout arrays
            orc = oracle_of(root_ctx if on_device else None, base)
            no = min(ns, 4000)
            d = orc.map_batch(sample[:no].reshape(-1), offs[:no + 1], w["tau"], threads=cpu_share())
            ho = ctx.map_batch(sample[:no].reshape(-1), offs[:no + 1], w["tau"])
            replica["oracle_sample_reads"] = no
            replica["identical_to_oracle"] = bool(np.array_equal(ho.status, d.status) and np.array_equal(ho.located(), d.located())
                                                  and ho.n_ext == d.n_ext)

    if rank == 0:
        total_reads = n_reads * world * args.steps
        value = total_reads / elapsed
        mean = {k: float(np.mean(v)) for k, v in per.items()}
        # dominant kernel: k_search.  Its units are the extension steps it executes on the Occ table (the rest of
        # the reference schedule's steps are served by the jump table and replayed on the text by k_verify).
        k_ms = mean["k_search"]
        occ_steps = c["n_ext"] - c["jumped_steps"] - c["verify_steps"]
        achieved = BYTES_PER_EXT * occ_steps / (k_ms * 1e-3) / 1e9
        traffic, traffic_src, requests = pmc_traffic(args.workload)
        gather = gather_ceiling(2 * (w["L"] + w["n_seqs"]) // 2)     # Occ blocks: half a byte per text position
        if gather:
            gather["algorithmic_Glines_per_s"] = 2.0 * occ_steps / (k_ms * 1e-3) / 1e9
            gather["frac_algorithmic"] = gather["algorithmic_Glines_per_s"] / gather["ceiling_Glines_per_s"]
            if requests:
                gather["hbm_requests_per_launch"] = requests
                gather["hbm_requests_Glines_per_s"] = requests / (k_ms * 1e-3) / 1e9
                gather["frac_of_gather"] = gather["hbm_requests_Glines_per_s"] / gather["ceiling_Glines_per_s"]
                gather["note"] = ("frac_of_gather = HBM read requests of k_search (committed PMC pass) / live kernel time / live ceiling; "
                                  "frac_algorithmic counts 2 lines per step and exceeds it where the two ranks of a step share a line")
        H, L, W = c["list_length"], c["n_locs"], e["verify_windows"]

        def kern(name, ms, units, unit, bytes_per_unit, note_):
            by = float(bytes_per_unit) * units
            return {"kernel": name, "ms": ms, "unit": unit, "units_per_launch": units, "algorithmic_bytes": by,
                    "achieved_GBps": by / (ms * 1e-3) / 1e9 if ms > 0 else None,
                    "frac_hbm": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None, "accounting": note_}
        kernels = [
            kern("k_search", k_ms, occ_steps, "extension step on the Occ table", 128, "two 64-B Occ lines per step (SURVEY 8d)"),
            kern("k_verify + k_verify_group", mean["k_verify"], W, "suffix replayed on the text", 128.0 + 80.0 * c["n_hits"] / max(W, 1),
                 "per suffix one 64-B line of the suffix array and one of the text's bit planes (SURVEY 8d: 64 B per random touch); per entry "
                 "of the search's list, read once, 32 B + 48 B of the query's planes; the counters of the committed PMC passes "
                 "(pmc_bytes, pmc_frac_hbm) are the check of this figure"),
            kern("k_stage_compact + scan + k_place + k_place_long", mean["locate"], L, "located position", 16,
                 "8-B key read from the blocks' staging slots and written to the dense array (the few hits that are intervals read "
                 "their suffix-array entries instead; not counted apart)"),
            kern("ordering (radix passes + k_seg_first + k_order_segments, or the full radix sort + k_read_first)", mean["order"], L,
                 "located position", 24,
                 "8-B key read and written once + CSR offset; the sort by segments moves the keys 3 times (2 radix passes over the "
                 "top 16 bits, then a workgroup orders each segment in LDS), the full sort 7 to 8 times"),
            kern("k_read_lengths + k_encode", mean["encode"], n_reads * w["m"], "letter",
                 1.0 + 0.375 * (128 * ((w["m"] + 127) // 128)) / w["m"],
                 "1 B letter in, 3 bits of bit planes out per position of the read's row (rows are multiples of 128 positions); "
                 "the byte matrix of the reads is not written in a mapping run"),
        ]
        vt = pmc_verify_traffic(args.workload)
        if vt:
            kernels[1]["pmc_bytes"] = vt["bytes"]
            kernels[1]["pmc_ms"] = vt["ms"]
            kernels[1]["pmc_frac_hbm"] = vt["bytes"] / (vt["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if vt["ms"] else None
            kernels[1]["pmc_source"] = vt["source"]
        out = {
            "metric": "reads mapped/sec (node), GRCh38-size index, %dnt <=%d mm; %% HBM gather roofline" % (w["m"], w["tau"])
                      if args.workload.startswith("grch38") else
                      "reads mapped/sec (node), %s index, %dnt <=%d mm; %% HBM gather roofline" % (args.workload, w["m"], w["tau"]),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": args.workload, "description": w["desc"], "text_len": int(2 * (w["L"] + w["n_seqs"])),
                       "reads_per_gpu_per_step": n_reads, "distinct_chunks_per_gpu": len(d_chunks), "read_len": w["m"], "tau": w["tau"],
                       "timed_region": "per step one chunk: letters resident in HBM -> encode + search + locate + order + D2H of "
                                       + ("packed results (5 B per hit = text position + score, in read order; 4-B offset + status byte per read)"
                                          if args.results == "packed" else "compact results (8-B keys + CSR offsets + status)") +
                                       " into page-locked host memory; "
                                       "chunks pipelined over %d batch objects (gpumap_pipe)" % args.depth,
                       "parallelism": "reads sharded over %d GPU(s), index replicated (RCCL broadcast)" % world},
            "located_per_step": located_total / args.steps,
            "step_intervals_ms": [round(1e3 * (b_ - a_), 2) for a_, b_ in zip([t0] + done_at[:-1], done_at)],
            "device_resident": {"reads_per_s": n_reads / (mean["device_total"] * 1e-3), "ms_per_chunk": mean["device_total"],
                                "note": "one chunk at a time on one batch object, results left in HBM (no D2H, no overlap): "
                                        "encode + search + locate + order; round 1's headline was this without the encoding"},
            "roofline": {"bound": "hbm", "kernel": "k_search", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": k_ms, "ext_steps_per_launch": occ_steps,
                         "algorithmic_bytes_per_launch": BYTES_PER_EXT * occ_steps,
                         "k_verify_ms": mean["k_verify"],
                         "gather": gather,
                         "kernels": kernels,
                         "reference_schedule": {"ext_steps": c["n_ext"], "bytes": BYTES_PER_EXT * c["n_ext"],
                                                "retired_GBps": BYTES_PER_EXT * c["n_ext"] / ((k_ms + mean["k_verify"]) * 1e-3) / 1e9,
                                                "note": "SURVEY 8d bytes of the whole reference schedule over k_search + k_verify "
                                                        "time: a rate of work retired, not a traffic (can exceed the HBM peak)"}},
            "stages_ms": {"encode": mean["encode"], "search": mean["k_search"] + mean["k_verify"], "locate": mean["locate"],
                          "order": mean["order"], "chunk_on_device": mean["device_total"]},
            "hits_per_batch": {"list_entries": c["n_hits"], "list_length": c["list_length"], "located": c["n_locs"],
                               "verify_windows": W, "big_candidates": e["big_candidates"]},
            # lane_fill counts the steps of the reference schedule only; a continuation (a node looked at again for its
            # next sibling) occupies its lane for a pass as well: lane_busy counts both
            "occupancy": {"loop_passes": c["loop_passes"], "continuations": c["continuations"],
                          "lane_busy": ((occ_steps + c["continuations"]) / (64.0 * c["loop_passes"])) if c["loop_passes"] else None,
                          "steps_from_jump_table": c["jumped_steps"], "steps_replayed_on_text": c["verify_steps"],
                          "steps_on_occ_table": occ_steps,
                          "lane_fill": (occ_steps / (64.0 * c["loop_passes"])) if c["loop_passes"] else None},
            "setup_s": {"index_build": t_index, "index_built_on": "device" if on_device else "host", "upload_repack": t_upload, "rccl_broadcast": t_bcast,
                        "replica_from_device": t_replica, "all": t_setup},
        }
        if replica:
            out["replica_check"] = replica
        if world == 1 and not args.no_pcie:
            out["pcie_inclusive"] = pcie_inclusive(ma, ctx, codes, offs, w, args.depth)
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(root_ctx if on_device else None, base, codes, offs, w, args.cpu_seconds, gpu_ctx=ctx)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    batch.close()
    pipe.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if root_ctx is not None and root_ctx is not ctx:
        root_ctx.close()
    del keep


def gather_ceiling(footprint_bytes):
    """The measured ceiling north_star asks for: dependent random 64-B line reads over a footprint of the Occ
    table's size, in the shape k_search uses (a quad of lanes per line), 32 waves per CU
    (tools/gather_bench.hip, run as a child process on the same GPU)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "gather_bench")
    if not os.path.exists(exe):
        return None
    mib = max(16, int(footprint_bytes) >> 20)
    try:
        r = subprocess.run([exe, "quick", str(mib), "32", "2000"], capture_output=True, text=True, timeout=120)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"ceiling_Glines_per_s": d["Gtouch_per_s"], "footprint_MiB": d["footprint_MiB"], "shape": d["shape"],
                "waves_per_cu": d["waves_per_cu"], "source": "tools/gather_bench.hip quick, this run"}
    except Exception as ex:
        return {"error": str(ex)}


def pcie_inclusive(ma, ctx, codes, offs, w, depth):
    """The boundary hands over host buffers: figures for DESIGN.md, never `value`.  (a) pipelined: chunks of letters
    in page-locked host memory through gpumap_pipe (H2D + encode + search + locate + order + D2H of packed results,
    the stages of consecutive chunks overlapping); (b) one shot: gpumap_map_batch, codes in, the full structure of
    arrays out, pageable (malloc'ed results, any caller) and page-locked (gpumap_host_alloc + option pinned_results)."""
    import ctypes as C
    lib = ma.lib()
    flat = np.ascontiguousarray(codes.reshape(-1))
    n = len(offs) - 1
    res = {}
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    pins = []
    for k in range(2):
        pin = lib.gpumap_host_alloc(flat.nbytes)
        v = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(flat.nbytes,))
        v[:] = lut[np.roll(flat, 1000 * k * w["m"])]                   # (the same reads in another order)
        pins.append((pin, v))
    pipe = ctx.pipe(depth=depth, tau=w["tau"], flags=ma.SRC_TEXT | ma.PIPE_PACKED)
    off_arr = np.ascontiguousarray(offs, dtype=np.int64)

    def go(k_steps):
        taken = 0
        for k in range(k_steps):
            if k - taken == depth:
                raw, _ = pipe.next_raw(); pipe.release_raw(raw); taken += 1
            if lib.gpumap_pipe_submit(pipe._h, C.c_void_p(pins[k % 2][0]), off_arr.ctypes.data_as(C.c_void_p), n, None) != 0:
                raise RuntimeError("gpumap_pipe_submit failed")
        while taken < k_steps:
            raw, _ = pipe.next_raw(); pipe.release_raw(raw); taken += 1
    go(depth)
    t0 = time.perf_counter()
    go(6)
    dt = (time.perf_counter() - t0) / 6
    pipe.close()
    res["pipelined_pinned"] = {"reads_per_s": n / dt, "seconds_per_chunk": dt, "chunks": 6,
                               "what": "letters in page-locked host memory -> H2D + encode + search + locate + order + D2H (compact), gpumap_pipe"}
    pflat = pins[0][1]
    pflat[:] = flat
    for mode, src in (("pageable", flat), ("pinned", pflat)):
        ctx.set_option("pinned_results", 1 if mode == "pinned" else 0)
        best = None
        for _ in range(2):
            raw = ma._Hits()
            t0 = time.perf_counter()
            rc = lib.gpumap_map_batch(ctx._h, src.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), len(offs) - 1,
                                      w["tau"], 0, C.byref(raw))
            dt = time.perf_counter() - t0
            assert rc == 0
            nl = int(raw.n_locs)
            lib.gpumap_hits_free(C.byref(raw))
            best = dt if best is None else min(best, dt)
        res[mode] = {"reads_per_s": (len(offs) - 1) / best, "seconds": best, "located": nl,
                     "what": "gpumap_map_batch: one shot, full structure of arrays"}
    ctx.set_option("pinned_results", 0)
    for pin, _ in pins:
        lib.gpumap_host_free(pin)
    return res


def pmc_traffic(workload):
    """HBM-side bytes and read requests per launch of the search kernel from the committed rocprofv3 PMC passes of
    this same command (tools/profile_passes.sh -> profiles/r<round>_<workload>_profile_summary.json): (FETCH_SIZE +
    WRITE_SIZE) * 1024.  FETCH_SIZE = TCC_EA0_RDREQ * 64 B here (every request of this kernel is one
    64-byte line, TCC_EA0_RDREQ_32B = 0), so the factor 2 of wide coalesced streams does not apply
    (MI355X_MICROARCH.md, HBM).  None when no profile of this workload is committed."""
    for rnd in ("r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", "%s_%s_profile_summary.json" % (rnd, workload))
        try:
            doc = json.load(open(path))
            pm = doc["pmc_per_launch"]
            k = next(v for name, v in pm.items() if "gm::k_search" in name)
            req = k.get("TCC_EA0_RDREQ_sum", {}).get("avg")
            src = "%s (counters of the build at commit %s)" % (os.path.relpath(path, ROOT), doc.get("commit", "unknown"))
            return (k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"]) * 1024.0, src, req
        except Exception:
            continue
    return None, None, None


def oracle_of(ctx, base):
    """The oracle (test infrastructure; here the checker and the CPU baseline) on the index files the host builder wrote, or
    on the reference-layout arrays of a device build copied back."""
    from oracle import pyoracle as po
    if ctx is None:
        return po.Index(base)
    ref = ctx.reference_arrays(want_text=False)
    return po.Index.from_arrays(ref["occ"], ref["c"], ref["sa"], ref["sar_bits"], ref["txt_len"], ref["wil_cnt"])


def cpu_share():
    try:
        share = len(os.sched_getaffinity(0))
    except Exception:
        share = os.cpu_count() or 1
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        share = min(share, int(os.environ["OMP_NUM_THREADS"]))          # the box's CPU share per GPU job
    return share


def pmc_verify_traffic(workload):
    """FETCH_SIZE + WRITE_SIZE and the kernel time of k_verify and k_verify_group per chunk from the committed PMC passes
    (tools/profile_passes.sh), or None."""
    for rnd in ("r3", "r2"):
        path = os.path.join(ROOT, "profiles", "%s_%s_profile_summary.json" % (rnd, workload))
        try:
            d = json.load(open(path))
            by, ms = 0.0, 0.0
            for name, v in d["pmc_per_launch"].items():
                if "gm::k_verify" in name:
                    by += (v["FETCH_SIZE"]["avg"] + v["WRITE_SIZE"]["avg"]) * 1024.0
            for v in d.get("kernel_stats_serial") or d["kernel_stats"]:
                if "gm::k_verify" in v["kernel"]:
                    ms += v["avg_ns"] * 1e-6                      # (one launch of each instance per chunk)
            if by > 0 and ms > 0:
                return {"bytes": by, "ms": ms, "source": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


def cpu_baseline(ctx, base, codes, offs, w, seconds, gpu_ctx=None):
    """The oracle's C restatement (kind "port") on the host cores, on a bounded sample of the same
    reads against the same index (the files the host builder wrote, or the reference-layout arrays of
    a device build copied back).  Test infrastructure used as the baseline only.  Three thread counts: 1, the CPUs
    this job may run on (its affinity mask / the box's share), and nproc (every CPU the machine shows)."""
    orc = oracle_of(ctx, base)
    nproc = os.cpu_count() or 1
    share = cpu_share()
    runs = {}
    best_t, best_d, best_n = None, None, 0
    for T in sorted(set([1, share, min(nproc, 512)])):
        probe = min(len(offs) - 1, 2000 * min(T, 64))
        d = orc.map_batch(codes[:probe].reshape(-1), offs[:probe + 1], w["tau"], threads=T)
        rate = probe / max(d.seconds, 1e-6)
        n = int(min(len(offs) - 1, max(probe, rate * seconds / 2)))
        d = orc.map_batch(codes[:n].reshape(-1), offs[:n + 1], w["tau"], threads=T)
        runs["T=%d" % T] = {"reads_per_s": n / d.seconds, "threads": T, "reads": n, "seconds": d.seconds}
        if best_d is None or n / d.seconds > best_n / best_d.seconds:
            best_t, best_d, best_n = T, d, n
    d, n = best_d, best_n
    out = {"value": n / d.seconds, "unit": "reads/s", "cores": best_t, "kind": "port", "nproc": nproc, "cpu_share": share,
           "sample": "first %d reads of the first chunk, %d pthreads, %.1f s (the fastest of the thread counts tried)" % (n, best_t, d.seconds),
           "by_threads": runs, "ext_steps_per_read": d.n_ext / n}
    if gpu_ctx is not None:
        # the same sample through the HIP path (compact results, as in the timed region): status, mapped positions,
        # strands (position >= n/2), scores and the count of extension steps must be identical
        gpu_ctx.set_option("compact_results", 1)
        b = gpu_ctx.upload(codes[:n].reshape(-1), offs[:n + 1])
        b.run(w["tau"])
        h = b.fetch()
        b.close()
        gpu_ctx.set_option("compact_results", 0)
        out["parity_on_sample"] = {"reads": n, "located": int(h.n_locs),
                                   "identical": bool(np.array_equal(h.status, d.status) and
                                                     np.array_equal(h.located(), d.located()) and h.n_ext == d.n_ext)}
    return out


if __name__ == "__main__":
    main()
