#!/bin/bash
# Full bench lines (CPU baseline included) of every workload, one process each: tools/bench_all.sh OUTDIR
out=$1; mkdir -p $out
for w in plumbing ecoli chr21 grch38_150 grch38; do
  timeout -k 10 500 python bench.py --workload $w > $out/${w}_bench.json 2> $out/${w}_bench.err || { echo "$w failed"; tail -5 $out/${w}_bench.err; exit 1; }
  echo "$w done"
done
