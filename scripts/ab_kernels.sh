#!/bin/bash
# GPU box: A/B the kernel versions of the default library on the bench workloads (kernel ms).
for kv in ${KERNELS:-1 3 4}; do
  RTC_KERNEL=$kv timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --extra-workloads config3,config3_high ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('kernel v$kv $RTC_ENV_NOTE: config2 %.2f ms (%.0f Mrays/s)  config3 %.3f ms  config3_high %.3f ms' % (j['roofline']['kernel_ms_avg'], j['value'], j['extra']['config3']['kernel_ms'], j['extra']['config3_high']['kernel_ms']))
"
done
