#!/bin/bash
# GPU box: per-launch durations of one wavefront frame (rocprofv3 --kernel-trace) + rays per level.  usage: scripts/wf_timeline.sh [workload]
export TMPDIR=/tmp
WL=${1:-config2}
OUT=gpurun_out/wf_timeline_$WL
rm -rf $OUT; mkdir -p $OUT
RTC_WF_DUMP_COUNTS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 scripts/diag_run.py $WL 4 > $OUT/run.log 2>&1
grep "rtc-wf" $OUT/run.log | tail -1
python3 - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]))
rows.sort()
wf = [r for r in rows if r[2].startswith(("wf_", "rtc_trace"))]
# the last frame: the last 14-ish wf launches ending with wf_gather
end = max(i for i, r in enumerate(wf) if r[2].startswith("wf_gather"))
start = max(i for i, r in enumerate(wf[:end]) if r[2].startswith("wf_gather")) + 1 if any(r[2].startswith("wf_gather") for r in wf[:end]) else 0
t0 = wf[start][0]
for s, e, n in wf[start:end + 1]:
    print("%8.1f us +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
print("frame: %.1f us" % ((wf[end][1] - t0) / 1e3))
PY
