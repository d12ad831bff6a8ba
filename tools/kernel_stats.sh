#!/bin/bash
# rocprofv3 kernel stats of one bench command: tools/kernel_stats.sh OUTDIR workload
out=$1; w=$2
R=$GRAFT_REPO_ROOT
raw=/tmp/gpumap_ks_$$
mkdir -p $R/$out $raw
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $raw/line.json 2> $raw/err.txt || { tail -5 $raw/err.txt; exit 1; }
cp "$(ls -S $(find $raw -name "*kernel_stats.csv") | head -1)" $R/$out/kernel_stats_$w.csv; cp $raw/line.json $R/$out/line_$w.json
python3 - $R/$out/kernel_stats_$w.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-90s calls %4s avg %10.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
