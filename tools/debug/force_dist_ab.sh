#!/bin/bash
out=gpurun_out/ab_fd; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --no-pcie --steps 12 > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; exit 1; }
python - $out/$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print(sys.argv[2], round(d["value"]/1e6,1), d["step_intervals_ms"])
PY
}
run default X=1
run noprio GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq16 GPU_MAX_HW_QUEUES=16
