#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench workload.
# usage: scripts/profile_gpu.sh <tag> [workload]     -> gpurun_out/prof_<tag>/...
# PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md §rocprofv3 PMC slots)
# and never combined with trace domains other than --kernel-trace.
set -u
TAG=${1:-r1}
WL=${2:-config2}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra-workloads= --workload $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- $BENCH > $OUT/pmc_$N.log 2>&1 || echo "pmc pass $N failed" >> $OUT/errors.txt
done
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
