#!/bin/bash
# rocprofv3 kernel stats of one gpumap_ann_build: tools/ann_kernel_stats.sh OUTDIR genome_len
out=$1; len=$2
R=$GRAFT_REPO_ROOT
raw=/tmp/gpumap_annks_$$
mkdir -p $R/$out $raw
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $raw -- python3 $R/tools/ann_bench.py --genome-len $len --device-build > $raw/line.json 2> $raw/err.txt || { tail -5 $raw/err.txt; exit 1; }
cp "$(ls -S $(find $raw -name "*kernel_stats.csv") | head -1)" $R/$out/ann_kernel_stats.csv; cp $raw/line.json $R/$out/ann_line.json
python3 - $R/$out/ann_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-100s calls %5s avg %10.1f us total %9.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
