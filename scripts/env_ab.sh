#!/bin/bash
# GPU box: the shipped library under two settings of one environment variable.  usage: scripts/env_ab.sh VAR "v1 v2 ..." [workload] [bench args]
VAR=$1; VALS=$2; WL=${3:-config2}; shift; shift; shift
mkdir -p gpurun_out
for v in $VALS; do
  env $VAR=$v timeout -k 10 200 python3 bench.py --workload $WL --steps 60 --warmup 3 --no-cpu-baseline --no-pmc --extra-workloads "" "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']; c = r['counters_rank0']
        print('$VAR=$v $WL: %.3f ms/step (F=%d)  sequential %.3f ms  path %s (1k %.3f / wf %.3f)  nodes %d analytic %d light cells %d' % (j['ms_per_step'], j['config']['frames_in_flight'], r['kernel_ms_avg'], r['path']['path'], r['path']['one_kernel_ms'], r['path']['wavefront_ms'], c['accel_nodes'], c['analytic_tests'], c.get('light_grid_cells', 0)))
"
done
