#!/bin/bash
# A/B over option settings of the library: bench lines of one workload per setting.
# usage: tools/opt_ab.sh OUTDIR workload "name=value name=value ..." (each setting one run; "-" = defaults)
out=$1; w=$2; shift; shift
mkdir -p $out
for o in "$@"; do
  tag=$(echo "$o" | tr ' =' '__')
  args=""
  if [ "$o" != "-" ]; then for x in $o; do args="$args --opt $x"; done; fi
  timeout -k 10 300 python bench.py --workload $w --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-pcie $args > $out/bench_${w}_$tag.json 2> $out/bench_${w}_$tag.err || { tail -5 $out/bench_${w}_$tag.err; exit 1; }
  python - $out/bench_${w}_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(sys.argv[1], "reads/s=%.4g ms/step=%.3f k_search=%.3f k_verify=%.3f located=%d" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["k_verify_ms"], d["hits_per_batch"]["located"]), {k: round(v, 3) for k, v in d["stages_ms"].items()})
PY
done
