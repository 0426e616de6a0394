#!/bin/bash
# GPU box: one rocprofv3 PMC pass of the default bench workload; prints per-dispatch means for the ray kernel.
# usage: scripts/pmc_pass.sh <tag> COUNTER [COUNTER...]      (PMC only with --kernel-trace, as the pool requires)
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
timeout -k 10 250 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --extra-workloads= ${BENCH_ARGS:-} > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rtc_trace_kernel<false' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print('%-28s %14.0f  (%d dispatches)' % (k, sum(v) / len(v), len(v)))
PY
