#!/bin/bash
# A/B over values of one environment variable: bench lines of one workload per value, STEPS steps each (default 20).
# usage: tools/env_ab.sh OUTDIR VAR "values" workload [extra bench args]
out=$1; var=$2; vals=$3; w=$4; shift; shift; shift; shift
mkdir -p $out
for v in $vals; do
  export $var=$v
  timeout -k 10 400 python bench.py --workload $w --steps ${STEPS:-20} --warmup 2 --no-cpu-baseline --no-pcie "$@" > $out/bench_${w}_$v.json 2> $out/bench_${w}_$v.err || { tail -5 $out/bench_${w}_$v.err; exit 1; }
  python - $out/bench_${w}_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(sys.argv[1], "reads/s=%.4g ms/step=%.3f k_search=%.3f k_verify=%.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["k_verify_ms"]), {k: round(v, 3) for k, v in d["stages_ms"].items()})
PY
done
