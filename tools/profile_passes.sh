#!/bin/bash
# rocprofv3 passes over one bench.py command (run on the GPU box):
#   tools/profile_passes.sh OUTDIR WORKLOAD [extra bench args]
# pass 0: --kernel-trace --stats; then one --pmc pass per counter group (never combined with traces).
# Each pass is its own process with its own time limit; the script stops at the first failure.
# pass 0 runs twice: as the product runs (chunks overlap on the pipeline's streams) and with GPUMAP_PIPE_OVERLAP=0
# (every launch alone on the device: the duration bench.py's HIP events measure).
# The counter passes run without stream priorities (the library's default since; set here in case the environment says
# otherwise): counter collection serialises the kernels of all queues, which makes a high-priority queue wait for the others,
# and a high-priority queue that waits for a lower one can stall this platform for good (seen: tests/test_gpu_mapper.py under
# --pmc with the priorities of before, 3 runs of 4; none without).
# PMC_ONLY="7 8" runs only the named counter passes (and no stats pass).  COMMIT=<id> is recorded in the summary.
# Raw rocprofv3 output stays in /tmp (gpurun_out/ is limited to 64 MiB); summaries go to OUTDIR.
out=$1; shift; wl=$1; shift
R=$GRAFT_REPO_ROOT
final=$R/$out
raw=/tmp/gpumap_prof_$$
mkdir -p $final $raw
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline $*"
if [ -z "$PMC_ONLY" ]; then
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -- $B > $raw/stats.json 2> $raw/stats.err || { echo "stats pass failed"; tail -5 $raw/stats.err; exit 1; }
GPUMAP_PIPE_OVERLAP=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats_serial -- $B > $raw/stats_serial.json 2> $raw/stats_serial.err || { echo "serial stats pass failed"; tail -5 $raw/stats_serial.err; exit 1; }
fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi
  GPUMAP_PIPE_PRIO=0 GPUMAP_PIPE_POST_PRIO=0 timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $raw/pmc$i -- $B > $raw/pmc$i.json 2> $raw/pmc$i.err || { echo "pmc pass $i ($grp) failed"; tail -5 $raw/pmc$i.err; exit 1; }
done
python3 $R/tools/profile_summary.py $raw > $final/summary.json || exit 1
[ -z "$PMC_ONLY" ] && cp "$(ls -S $(find $raw/stats -name "*kernel_stats.csv") | head -1)" $final/kernel_stats.csv   # (the largest: the bench, not the gather microbenchmark it starts)
[ -z "$PMC_ONLY" ] && cp "$(ls -S $(find $raw/stats_serial -name "*kernel_stats.csv") | head -1)" $final/kernel_stats_serial.csv
cp $raw/*.json $final/ 2>/dev/null
for d in $raw/pmc*/; do n=$(basename $d); head -1 $d/*/*counter_collection.csv > $final/$n.counters.csv; grep -E "k_search|k_verify|k_place|k_encode" $d/*/*counter_collection.csv >> $final/$n.counters.csv; done
cat $final/summary.json
