#!/bin/bash
# repeats of the pipeline's timed region with / without stream priorities, with / without the N > 1 path's process group
out=gpurun_out/ab_prio4; mkdir -p $out
run() { w=$1; tag=$2; extra=$3; shift; shift; shift; env "$@" timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-pcie --steps 12 $extra > $out/${w}_$tag.json 2> $out/${w}_$tag.err || { tail -3 $out/${w}_$tag.err; exit 1; }
python - $out/${w}_$tag.json $w $tag <<'PY'
import json,sys,statistics
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
iv=d["step_intervals_ms"]
print(sys.argv[2], sys.argv[3], "value %.1f M" % (d["value"]/1e6), "first %.1f median %.2f max-after-first %.1f" % (iv[0], statistics.median(iv[1:-1]), max(iv[1:])))
PY
}
for rep in 1 2 3; do
run grch38 plain_prio00_$rep "" GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run grch38 dist_prio00_$rep --force-dist GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run grch38 plain_prio11_$rep "" GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=1
run grch38 dist_prio11_$rep --force-dist GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=1
done
run chr21 plain_prio00 "" GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run chr21 dist_prio00 --force-dist GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run chr21 plain_prio11 "" GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=1
run chr21 dist_prio11 --force-dist GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=1
