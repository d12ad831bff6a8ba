#!/usr/bin/env python3
"""Hits per read of one chunk of a bench workload (what the ordering stage has to put in order): tools/hits_hist.py [workload]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mapper_amd as ma
from mapper_amd import synth, synth_device
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "grch38"
w = bench.WORKLOADS[name]
if w.get("device"):
    g, lens, names = synth_device.make_genome(w["L"], w["seed"], n_seqs=w["n_seqs"], gc=w["gc"], repeat_rich=w["rich"])
    ctx = ma.Context.build_device(lens, 0, keep_reference=False, device_ptr=g.data_ptr())
else:
    seqs, names = synth.make_genome(w["L"], w["seed"], n_seqs=w["n_seqs"], gc=w["gc"], repeat_rich=w["rich"])
    ctx = ma.Context.from_index(ma.HostIndex.build_codes(seqs, names, "/tmp/hh_" + name), 0)
    g = torch.from_numpy(np.concatenate(seqs)).cuda()
    lens = np.array([len(x) for x in seqs], dtype=np.int64)
n = min(w["reads"], 4_000_000)
rd = synth_device.make_reads(g, lens, n, w["m"], w["tau"], seed=w["seed"] * 1000 + 17).cpu().numpy()
ctx.set_option("compact_results", 1)
h = ctx.map_batch(rd.reshape(-1), np.arange(n + 1, dtype=np.int64) * w["m"], w["tau"])
per = np.diff(h.read_first_loc)
edges = [0, 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 1025, 4097, 1 << 30]
out = {"workload": name, "reads": int(n), "hits": int(per.sum()), "max": int(per.max()), "classes": {}}
for a, b in zip(edges[:-1], edges[1:]):
    m = (per >= a) & (per < b)
    out["classes"]["%d..%d" % (a, b - 1)] = {"reads": int(m.sum()), "hits": int(per[m].sum()), "sum_len_sq": int((per[m].astype(np.int64) ** 2).sum())}
print(json.dumps(out))
