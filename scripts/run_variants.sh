#!/bin/bash
# GPU box: bench each csrc/variants/*.so briefly (kernel ms for config2 / config3 / config3_high) -> gpurun_out/variants.txt
mkdir -p gpurun_out
: > gpurun_out/variants.txt
for so in raytracer_challenge_amd/csrc/variants/*.so; do
  RTC_AMD_LIB=$PWD/$so timeout -k 10 120 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --extra-workloads config3,config3_high 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%-28s config2 %.2f ms  config3 %.3f ms  config3_high %.3f ms' % ('$(basename $so)', j['roofline']['kernel_ms_avg'], j['extra']['config3']['kernel_ms'], j['extra']['config3_high']['kernel_ms']))
" >> gpurun_out/variants.txt
done
cat gpurun_out/variants.txt
