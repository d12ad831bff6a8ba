#!/bin/bash
# sweep of option verify_tile at one workload: tools/tile_sweep.sh OUTDIR WORKLOAD "tiles" [extra bench args]
out=$1; shift; w=$1; shift; tiles=$1; shift
mkdir -p $out
for t in $tiles; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-pcie --verify-tile $t "$@" > $out/tile_$t.json 2> $out/tile_$t.err || { tail -5 $out/tile_$t.err; exit 1; }
  python - $out/tile_$t.json $t <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print("tile", sys.argv[2], "reads/s=%.4g ms/step=%.2f k_search=%.2f k_verify=%.2f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["k_verify_ms"]), d["stages_ms"])
PY
done
