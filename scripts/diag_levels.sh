#!/bin/bash
# GPU box, RTC_DIAG variant built as `scripts/build_variant.sh diag -DRTC_DIAG`: section utilisation of one frame at several fuels (the
# difference between two fuels = what the deeper levels add).  usage: scripts/diag_levels.sh [workload] [kernel] [fuels...]
WL=${1:-config2}; K=${2:-4}; shift; shift
FUELS=${*:-0 1 2 3 5}
export RTC_AMD_LIB=$PWD/raytracer_challenge_amd/csrc/variants/librtc_amd_diag.so RTC_DIAG_DUMP=1
for f in $FUELS; do
  echo "== $WL kernel $K fuel $f"
  timeout -k 10 120 python3 scripts/diag_run.py $WL $K $f > /tmp/diag_$f.log 2>&1 || { tail -5 /tmp/diag_$f.log; exit 1; }
  grep kernel_ms /tmp/diag_$f.log
  python3 scripts/diag_report.py < /tmp/diag_$f.log
done
