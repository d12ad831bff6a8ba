import json, sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "value %.1f M" % (d["value"]/1e6), "ms/step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["stages_ms"].items()}, "frac", round(d["roofline"]["frac"],3))
    for k in d["roofline"]["kernels"]: print("   ", k["kernel"], round(k["ms"],2), round(k["frac_hbm"],3))
    if "cpu_baseline" in d: print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("parity_on_sample"))
