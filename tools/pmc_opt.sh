#!/bin/bash
# One rocprofv3 counter pass of a bench.py run with option settings, summed per kernel:
# tools/pmc_opt.sh OUTDIR workload "COUNTER ..." "kernel pattern" [bench args]
out=$1; w=$2; ctrs=$3; pat=$4; shift; shift; shift; shift
R=$GRAFT_REPO_ROOT; raw=/tmp/gpumap_pmo_$$
mkdir -p $R/$out $raw
cd /tmp && export TMPDIR=/tmp
GPUMAP_PIPE_PRIO=0 timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $raw -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-pcie "$@" > $raw/line.json 2> $raw/err.txt || { tail -5 $raw/err.txt; exit 1; }
f="$(ls -S $(find $raw -name "*counter_collection.csv") | head -1)"
python3 - "$f" "$pat" <<'PY'
import csv,sys,re,collections
tot=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        k=(r["Kernel_Name"][:50], r["Counter_Name"]); tot[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(tot): print("%-50s %-28s launches %3d per launch %.4g" % (k[0], k[1], len(n[k]), tot[k]/len(n[k])))
PY
rm -rf $raw
