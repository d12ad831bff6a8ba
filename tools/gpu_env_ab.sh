#!/bin/bash
# A/B over values of one environment variable: bench lines of one workload per value.
# usage: tools/gpu_env_ab.sh OUTDIR VAR "values" workload
out=$1; var=$2; vals=$3; w=$4
mkdir -p $out
for v in $vals; do
  export $var=$v
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_${w}_$v.json 2> $out/bench_${w}_$v.err || { tail -5 $out/bench_${w}_$v.err; exit 1; }
  python - $out/bench_${w}_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]; o=d["occupancy"]
print(sys.argv[1], "reads/s=%.3g k_search=%.2f ms k_verify=%.2f ms" % (d["value"], r["kernel_ms"], r["k_verify_ms"]),
      "steps occ/jump/text = %.3g/%.3g/%.3g" % (o["steps_on_occ_table"], o["steps_from_jump_table"], o["steps_replayed_on_text"]), d["stages_ms"])
PY
done
