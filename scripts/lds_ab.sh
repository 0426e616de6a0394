#!/bin/bash
# GPU box: both device paths with and without the LDS-resident scene tables (RTC_WF_LDS=0 / RTC_K1_LDS=0) on one library
for k in 4 1; do
  for v in 1 0; do
    for wl in ${WLS:-config2 config3}; do
      RTC_KERNEL=$k RTC_WF_LDS=$v RTC_K1_LDS=$v timeout -k 10 150 python3 bench.py --workload $wl --steps 60 --warmup 3 --inflight 1 --no-cpu-baseline --no-pmc --extra-workloads "" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('path %s  LDS tables %s  %-14s %.3f ms/step  sequential %.3f ms' % ('wavefront' if '$k' == '4' else 'one kernel', 'on ' if '$v' == '1' else 'off', '$wl', j['ms_per_step'], r['kernel_ms_avg']))
"
    done
  done
done
