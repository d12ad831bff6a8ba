#!/bin/bash
# A/B of two builds of the library: bench lines of the named workloads with each (GPUMAP_LIB names the build).
# usage: tools/lib_ab.sh OUTDIR "libA libB" "workloads" [extra bench args]
out=$1; shift; libs=$1; shift; wls=$1; shift
mkdir -p $out
for w in $wls; do
  for l in $libs; do
    tag=$(basename $(dirname $l))
    GPUMAP_LIB=$GRAFT_REPO_ROOT/$l timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-pcie "$@" > $out/bench_${w}_$tag.json 2> $out/bench_${w}_$tag.err || { tail -5 $out/bench_${w}_$tag.err; exit 1; }
    python - $out/bench_${w}_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]; o=d["occupancy"]
print(sys.argv[1], "reads/s=%.4g k_search=%.3f ms k_verify=%.3f ms" % (d["value"], r["kernel_ms"], r["k_verify_ms"]),
      "steps occ/jump/text = %.4g/%.4g/%.4g located=%d" % (o["steps_on_occ_table"], o["steps_from_jump_table"], o["steps_replayed_on_text"], d["hits_per_batch"]["located"]), d["stages_ms"])
PY
  done
done
