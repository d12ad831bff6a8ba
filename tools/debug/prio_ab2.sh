#!/bin/bash
out=gpurun_out/ab_prio2; mkdir -p $out; w=grch38
run() { tag=$1; shift; extra=$1; shift; env "$@" timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-pcie --steps 12 $extra > $out/${w}_$tag.json 2> $out/${w}_$tag.err || { tail -3 $out/${w}_$tag.err; exit 1; }
python - $out/${w}_$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print(sys.argv[2], round(d["value"]/1e6,1), d["step_intervals_ms"])
PY
}
run dist_prio10 --force-dist GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=0
run dist_prio01 --force-dist GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=1
run dist_hwq8_prio10 --force-dist GPU_MAX_HW_QUEUES=8 GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=0
run dist_hwq8_prio00 --force-dist GPU_MAX_HW_QUEUES=8 GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run hwq8_prio11 "" GPU_MAX_HW_QUEUES=8 GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=1
run hwq8_prio10 "" GPU_MAX_HW_QUEUES=8 GPUMAP_PIPE_PRIO=1 GPUMAP_PIPE_POST_PRIO=0
run hwq8_prio00 "" GPU_MAX_HW_QUEUES=8 GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
