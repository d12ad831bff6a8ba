#!/bin/bash
# rocprofv3 kernel stats of one bench.py run with option settings: tools/kernel_stats_opt.sh OUTDIR workload "pattern" [bench args]
out=$1; w=$2; pat=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT; raw=/tmp/gpumap_kso_$$
mkdir -p $R/$out $raw
cd /tmp && export TMPDIR=/tmp
GPUMAP_PIPE_OVERLAP=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-pcie "$@" > $raw/line.json 2> $raw/err.txt || { tail -5 $raw/err.txt; exit 1; }
f="$(ls -S $(find $raw -name "*kernel_stats.csv") | head -1)"
cp "$f" $R/$out/kernel_stats.csv
python3 - "$f" "$pat" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]): print("%-70s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
