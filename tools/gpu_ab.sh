#!/bin/bash
# A/B helper for GPU runs: parity tests, then bench lines for the named workloads.
# usage: tools/gpu_ab.sh OUTDIR "workloads" [extra bench args]
out=$1; shift; wls=$1; shift
mkdir -p $out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest.log 2>&1
rc=$?; tail -3 $out/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for w in $wls; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
  python - $out/bench_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]; o=d["occupancy"]
print(sys.argv[1], "reads/s=%.3g k_search=%.2f ms k_verify=%.2f ms frac=%.3f" % (d["value"], r["kernel_ms"], r["k_verify_ms"], r["frac"]),
      "steps occ/jump/text = %.3g/%.3g/%.3g fill=%.2f" % (o["steps_on_occ_table"], o["steps_from_jump_table"], o["steps_replayed_on_text"], o["lane_fill"] or 0), d["stages_ms"])
PY
done
