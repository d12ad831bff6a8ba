#!/bin/bash
# Timeline of the pipelined bench under rocprofv3 (kernel + copy trace): tools/timeline.sh OUTFILE workload [window_ms] [min_us] [bench args]
out=$GRAFT_REPO_ROOT/$1; w=$2; win=${3:-60}; minus=${4:-100}; shift; shift; shift; shift
raw=/tmp/gpumap_tl_$$
mkdir -p $(dirname $out) $raw
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $raw -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-pcie "$@" > $raw/line.json 2> $raw/err.txt || { tail -5 $raw/err.txt; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pipe_timeline.py $raw $win $minus > $out
rm -rf $raw
