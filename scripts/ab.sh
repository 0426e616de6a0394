#!/bin/bash
# GPU box: bench each csrc/variants/*.so briefly -> gpurun_out/ab.txt   (usage: scripts/ab.sh [workload] [extra bench args])
mkdir -p gpurun_out
WL=${1:-config2}; shift
: > gpurun_out/ab.txt
for so in raytracer_challenge_amd/csrc/variants/*.so; do
  RTC_AMD_LIB=$PWD/$so timeout -k 10 150 python3 bench.py --workload $WL --steps 60 --warmup 3 --no-cpu-baseline --no-pmc --extra-workloads "" "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        r = j['roofline']
        print('%-34s $WL: %.3f ms/step (F=%d)  sequential %.3f ms  path %s (1k %.3f / wf %.3f)' % ('$(basename $so)', j['ms_per_step'], j['config']['frames_in_flight'], r['kernel_ms_avg'], r['path']['path'], r['path']['one_kernel_ms'], r['path']['wavefront_ms']))
" >> gpurun_out/ab.txt
done
cat gpurun_out/ab.txt
