#!/bin/bash
out=gpurun_out/ab_prio3; mkdir -p $out; w=${W:-grch38}
run() { tag=$1; shift; extra=$1; shift; env GPUMAP_PIPE_DEBUG=1 "$@" timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-pcie --steps 12 $extra > $out/${w}_$tag.json 2> $out/${w}_$tag.err || { tail -3 $out/${w}_$tag.err; exit 1; }
grep "^\[pipe\] .* stream:" $out/${w}_$tag.err | sort | uniq -c | head -4
python - $out/${w}_$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print(sys.argv[2], round(d["value"]/1e6,1), d["step_intervals_ms"])
PY
}
run plain "" X=1
run dist --force-dist X=1
run plain_prio00 "" GPUMAP_PIPE_PRIO=0
run dist_prio00 --force-dist GPUMAP_PIPE_PRIO=0
run plain_prio11 "" GPUMAP_PIPE_POST_PRIO=1
run dist_prio11 --force-dist GPUMAP_PIPE_POST_PRIO=1
run plain_noprobe "" GPUMAP_PIPE_PROBE=0
