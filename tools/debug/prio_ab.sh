#!/bin/bash
# stream priorities of the pipeline on / off, with and without the N > 1 path's process group: tools/debug/prio_ab.sh WORKLOAD
out=gpurun_out/ab_prio; mkdir -p $out; w=${1:-grch38}
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-pcie --steps 20 $EXTRA > $out/${w}_$tag.json 2> $out/${w}_$tag.err || { tail -3 $out/${w}_$tag.err; exit 1; }
python - $out/${w}_$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print(sys.argv[2], round(d["value"]/1e6,1), d["step_intervals_ms"])
PY
}
run prio11 GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=1
run prio00 GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run prio10 GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=0
run prio01 GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=1
run prio00b GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
