#!/bin/bash
# bench lines of one workload over "ENV=val,--flag val" settings: tools/gpu_sweep.sh OUTDIR workload "setting;setting;..."
out=$1; w=$2; IFS=';' read -ra sets <<< "$3"
mkdir -p $out
i=0
for s in "${sets[@]}"; do
  i=$((i+1))
  envs=$(echo "$s" | cut -d, -f1); flags=$(echo "$s" | cut -d, -f2-)
  [ "$flags" == "$s" ] && flags=""
  env $envs timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline $flags > $out/b$i.json 2> $out/b$i.err || { tail -5 $out/b$i.err; exit 1; }
  python - $out/b$i.json "$s" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%-40s reads/s=%.3g k_search=%.2f k_verify=%.2f total=%.2f" % (sys.argv[2], d["value"], r["kernel_ms"], r["k_verify_ms"], d["stages_ms"]["total"]))
PY
done
