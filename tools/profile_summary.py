#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_passes.sh into one JSON document:
kernel-trace stats of our kernels and, per PMC counter, the average per launch of k_search and of k_verify."""
import csv, glob, json, os, sys

def main(root):
    out = {"commit": os.environ.get("COMMIT", "unknown"), "kernel_stats": [], "kernel_stats_serial": [],
           "pmc_per_launch": {}, "bench_lines": {}}
    # "stats": the product configuration (chunks of the pipeline overlap on four streams, so a kernel's launches beside
    # another chunk's ordering take longer); "stats_serial": GPUMAP_PIPE_OVERLAP=0, every launch alone on the device,
    # the figure the HIP events of bench.py's roofline measure.
    for sub, key in (("stats", "kernel_stats"), ("stats_serial", "kernel_stats_serial")):
        for f in glob.glob(os.path.join(root, sub, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Name"]
                if "gm::" in name or "rocprim" in name or "rocclr" in name:
                    out[key].append({"kernel": name.split("(")[0][:90], "calls": int(r["Calls"]),
                                     "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                     "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])})
    for f in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        acc = {}
        for r in csv.DictReader(open(f)):
            if not any(k in r["Kernel_Name"] for k in ("k_search", "k_verify", "k_place", "k_encode", "k_read_first", "Onesweep", "onesweep")):
                continue
            key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            d = acc.setdefault(key, {})
            d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        for (k, c), d in acc.items():
            vals = list(d.values())
            out["pmc_per_launch"].setdefault(k, {})[c] = {"launches": len(vals), "avg": sum(vals) / len(vals)}
    for f in sorted(glob.glob(os.path.join(root, "*.json"))):
        if os.path.basename(f) == "summary.json":
            continue
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            out["bench_lines"][os.path.basename(f)] = {"value": d["value"], "kernel_ms": d["roofline"]["kernel_ms"],
                                                       "k_verify_ms": d["roofline"].get("k_verify_ms"),
                                                       "ext_steps": d["roofline"]["ext_steps_per_launch"]}
        except Exception as e:  # a pass without a bench line
            out["bench_lines"][os.path.basename(f)] = str(e)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(sys.argv[1])
