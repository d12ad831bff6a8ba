#!/usr/bin/env python3
"""bench.py -- throughput of the query hot path (block search + locate) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One process per GPU (for N > 1 launched by torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* come from the environment).  Rank 0 builds the synthetic index on the host, uploads and
repacks it on its GPU and broadcasts the three device arrays once over xGMI with RCCL
(torch.distributed, backend "nccl"); every rank then maps its own shard of reads against its own
replica.  No collective on the data path: reads are independent units (weak scaling: the per-GPU
batch is fixed, value = reads of all ranks / max-over-ranks time).

A "step" is one pass of the hot path over one batch of reads already resident in HBM:
gpumap_batch_run = search (k_search) + suffix-array locate + ordering, results left on the device.
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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("GPUMAP_WORKLOAD", "grch38"))
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--verify", type=int, default=-1, help="option 'verify' of the library (-1 = its default by text size)")
    ap.add_argument("--device-build", action="store_true", help="generate and index in HBM also for the small workloads")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 "
                    "on a box with fewer GPUs than ranks: ranks then share devices)")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    t_index = t_upload = t_bcast = 0.0
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
            ctx = ma.Context.build_device(lens, local, keep_reference=want_cpu, device_ptr=g.data_ptr())
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
    if world > 1:
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
        if rank != 0:
            d = ma.DevIndex(bufs[0].data_ptr(), meta[1], bufs[1].data_ptr(), meta[2], bufs[2].data_ptr(), meta[3],
                            bufs[3].data_ptr() if bufs[3] is not None else None, meta[4])
            ctx = ma.Context.from_device(d, meta[0], local, keep=bufs)
        keep = bufs
        t_bcast = time.time() - t0
    if args.waves_per_cu:
        ctx.set_option("waves_per_cu", args.waves_per_cu)
    if args.verify >= 0:
        ctx.set_option("verify", args.verify)

    # this rank's shard: its own reads (weak scaling), resident in HBM before the timed region
    if on_device:
        rd = synth_device.make_reads(g, lens, w["reads"], w["m"], w["tau"], seed=read_seed)
        codes = rd.cpu().numpy()
        offs = np.arange(w["reads"] + 1, dtype=np.int64) * w["m"]
        del rd, g
        torch.cuda.empty_cache()
    else:
        codes, offs = synth.make_reads(seqs, w["reads"], w["m"], w["tau"], seed=read_seed)
    batch = ctx.upload(codes.reshape(-1), offs)
    t_setup = time.time() - t_setup
    note("reads resident, timing")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run(w["tau"])
    barrier()
    t0 = time.perf_counter()
    ms_search, ms_k_search, ms_k_verify = [], [], []
    for _ in range(args.steps):
        c = batch.run(w["tau"])                      # synchronous: returns when the device work is done
        ms_search.append(c["ms_search"])
        ms_k_search.append(c["us_k_search"] / 1000.0)
        ms_k_verify.append(c["us_k_verify"] / 1000.0)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    c = batch.counters()

    if rank == 0:
        total_reads = w["reads"] * world * args.steps
        value = total_reads / elapsed
        # dominant kernel: k_search.  Its units are the extension steps it executes on the Occ table (the rest of
        # the reference schedule's steps are served by the jump table and replayed on the text by k_verify).
        k_ms = float(np.mean(ms_k_search))
        occ_steps = c["n_ext"] - c["jumped_steps"] - c["verify_steps"]
        achieved = BYTES_PER_EXT * occ_steps / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload)
        search_ms = float(np.mean(ms_search))
        out = {
            "metric": "reads mapped/sec (node), 100nt <=2 mm; % HBM gather roofline",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": args.workload, "description": w["desc"], "text_len": int(2 * (w["L"] + w["n_seqs"])),
                       "reads_per_gpu": w["reads"], "read_len": w["m"], "tau": w["tau"],
                       "parallelism": "reads sharded over %d GPU(s), index replicated (RCCL broadcast)" % world},
            "roofline": {"bound": "hbm", "kernel": "k_search", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": k_ms, "ext_steps_per_launch": occ_steps,
                         "algorithmic_bytes_per_launch": BYTES_PER_EXT * occ_steps,
                         "k_verify_ms": float(np.mean(ms_k_verify)),
                         "reference_schedule": {"ext_steps": c["n_ext"], "bytes": BYTES_PER_EXT * c["n_ext"],
                                                "retired_GBps": BYTES_PER_EXT * c["n_ext"] / (search_ms * 1e-3) / 1e9,
                                                "note": "SURVEY 8d bytes of the whole reference schedule over k_search + k_verify "
                                                        "time: a rate of work retired, not a traffic (can exceed the HBM peak)"}},
            "stages_ms": {"search": c["ms_search"], "locate": c["ms_locate"], "order": c["ms_sort"],
                          "total": c["ms_total"]},
            "hits_per_batch": {"list_entries": c["n_hits"], "list_length": c["list_length"], "located": c["n_locs"]},
            "occupancy": {"loop_passes": c["loop_passes"], "continuations": c["continuations"],
                          "steps_from_jump_table": c["jumped_steps"], "steps_replayed_on_text": c["verify_steps"],
                          "steps_on_occ_table": c["n_ext"] - c["jumped_steps"] - c["verify_steps"],
                          "lane_fill": ((c["n_ext"] - c["jumped_steps"] - c["verify_steps"]) / (64.0 * c["loop_passes"]))
                          if c["loop_passes"] else None},
            "setup_s": {"index_build": t_index, "index_built_on": "device" if on_device else "host", "upload_repack": t_upload, "rccl_broadcast": t_bcast,
                        "all": t_setup},
        }
        if world == 1:
            out["pcie_inclusive"] = pcie_inclusive(ma, ctx, codes, offs, w)
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(ctx if on_device else None, base, codes, offs, w, args.cpu_seconds, gpu_ctx=ctx)
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    del keep


def pcie_inclusive(ma, ctx, codes, offs, w):
    """The boundary hands over host buffers: untimed-region figure for DESIGN.md, never `value`.  One pass of
    upload (H2D of the reads) + search + locate + order + fetch (D2H of every located hit) per buffer kind:
    pageable (malloc'ed results, any caller) and page-locked (gpumap_host_alloc + option pinned_results)."""
    import ctypes as C
    lib = ma.lib()
    flat = np.ascontiguousarray(codes.reshape(-1))
    res = {}
    pin = lib.gpumap_host_alloc(flat.nbytes)
    pflat = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(flat.nbytes,))
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
        res[mode] = {"reads_per_s": (len(offs) - 1) / best, "seconds": best, "located": nl}
    ctx.set_option("pinned_results", 0)
    lib.gpumap_host_free(pin)
    return res


def pmc_traffic(workload):
    """HBM-side bytes per launch of the search kernel from the committed rocprofv3 PMC passes of this same
    command (tools/profile_passes.sh -> profiles/r1_<workload>_profile_summary.json): (FETCH_SIZE +
    WRITE_SIZE) * 1024.  FETCH_SIZE = TCC_EA0_RDREQ * 64 B here (every request of this kernel is one
    64-byte line, TCC_EA0_RDREQ_32B = 0), so the factor 2 of wide coalesced streams does not apply
    (MI355X_MICROARCH.md, HBM).  None when no profile of this workload is committed."""
    path = os.path.join(ROOT, "profiles", "r1_%s_profile_summary.json" % workload)
    try:
        pm = json.load(open(path))["pmc_per_launch"]
        k = pm.get("gm::k_search") or pm.get("gm::k_search_lane") or next(iter(pm.values()))
        return (k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"]) * 1024.0, os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def cpu_baseline(ctx, base, codes, offs, w, seconds, gpu_ctx=None):
    """The oracle's C restatement (kind "port") on the host cores, on a bounded sample of the same
    reads against the same index (the files the host builder wrote, or the reference-layout arrays of
    a device build copied back).  Test infrastructure used as the baseline only."""
    from oracle import pyoracle as po
    if ctx is None:
        orc = po.Index(base)
    else:
        ref = ctx.reference_arrays(want_text=False)
        orc = po.Index.from_arrays(ref["occ"], ref["c"], ref["sa"], ref["sar_bits"], ref["txt_len"], ref["wil_cnt"])
    cores = min(16, os.cpu_count() or 1)
    probe = min(len(offs) - 1, 4000 * cores)
    d = orc.map_batch(codes[:probe].reshape(-1), offs[:probe + 1], w["tau"], threads=cores)
    rate = probe / max(d.seconds, 1e-6)
    n = int(min(len(offs) - 1, max(probe, rate * seconds)))
    d = orc.map_batch(codes[:n].reshape(-1), offs[:n + 1], w["tau"], threads=cores)
    out = {"value": n / d.seconds, "unit": "reads/s", "cores": cores, "kind": "port",
           "sample": "first %d reads of the batch, %d pthreads, %.1f s" % (n, cores, d.seconds),
           "ext_steps_per_read": d.n_ext / n}
    if gpu_ctx is not None:
        # the same sample through the HIP path: mapped positions, strands (position >= n/2) and scores must be identical
        h = gpu_ctx.map_batch(codes[:n].reshape(-1), offs[:n + 1], w["tau"])
        out["parity_on_sample"] = {"reads": n, "located": int(h.n_locs),
                                   "identical": bool(np.array_equal(h.status, d.status) and
                                                     np.array_equal(h.located(), d.located()) and h.n_ext == d.n_ext)}
    return out


if __name__ == "__main__":
    main()
