#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (counters alone, never with a trace domain besides --kernel-trace) of a short bench run, per-kernel means.
# usage: scripts/pmc_counters.sh <tag> "<COUNTER ...>" [bench args]
export TMPDIR=/tmp
TAG=$1; CNT=$2; shift; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline --no-pmc --extra-workloads "" "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 scripts/pmc_by_kernel.py $OUT | grep -E "^#|wf_ts<false|wf_shade<false|wf_gather|rtc_trace_kernel<false"
