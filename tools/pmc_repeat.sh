#!/bin/bash
# The same counter pass of one workload N times (each its own process, stops at the first failure):
# looking for a run in which bench.py's own checks fail.   usage: tools/pmc_repeat.sh OUTDIR workload N
out=$GRAFT_REPO_ROOT/$1; w=$2; n=$3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $n); do
  timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcrep_$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $out/rep_$i.json 2> $out/rep_$i.err
  rc=$?
  echo "run $i rc=$rc $(grep -o '"located": [0-9]*' $out/rep_$i.json) $(grep 'bench.py:' $out/rep_$i.err)"
  rm -rf /tmp/pmcrep_$i
  if [ $rc -ne 0 ]; then tail -5 $out/rep_$i.err; exit 1; fi
done
