#!/bin/bash
# GPU box: throughput of the default workloads against the number of frames in flight per GPU
for wl in config2 config3; do
  for f in ${FS:-1 2 3 4 6}; do
    timeout -k 10 150 python3 bench.py --workload $wl --inflight $f --steps 240 --no-pmc --no-cpu-baseline --extra-workloads "" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$wl  frames in flight $f: %.3f ms/step  %.0f Mrays/s' % (j['ms_per_step'], j['value']))
"
  done
done
