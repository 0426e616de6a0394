#!/usr/bin/env python3
"""Condenses a scripts/profile_gpu.sh output directory into a short text summary (the file committed under profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("# profile summary for", out)
for f in sorted(glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
    print("\n## rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
    print(open(f).read().strip())
log = os.path.join(out, "trace.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            j = json.loads(line)
            print("\n## bench line of the profiled run (profiled clocks; not a headline number)")
            print(json.dumps({k: j[k] for k in ("value", "unit", "ms_per_step", "roofline")}))
print("\n## PMC passes: one-kernel path = per-dispatch mean over rtc_trace_kernel<false>; wavefront path = per-FRAME sum over the")
print("## wf_ts<false> / wf_shade<false> / wf_gather dispatches (frames = #wf_ts<false> - #wf_shade<false>); counting variants <true> excluded")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        acc, n = defaultdict(float), defaultdict(int)
        wf, n_ts, n_shade, n_gather_all, n_count_frames = defaultdict(float), defaultdict(int), defaultdict(int), defaultdict(int), defaultdict(int)
        for row in csv.DictReader(open(f)):
            kn, cn, v = row.get("Kernel_Name", ""), row["Counter_Name"], float(row["Counter_Value"])
            if "rtc_trace_kernel<false" in kn:
                acc[cn] += v
                n[cn] += 1
            elif "wf_ts<false" in kn or "wf_shade<false" in kn:
                wf[cn] += v
                if "wf_ts<false" in kn:
                    n_ts[cn] += 1
                else:
                    n_shade[cn] += 1
        for k in sorted(acc):
            print("%-32s one-kernel mean/dispatch = %.6g   (dispatches %d)" % (k, acc[k] / n[k], n[k]))
        for k in sorted(wf):
            frames = max(1, n_ts[k] - n_shade[k])
            print("%-32s wavefront sum/frame (wf_ts + wf_shade) = %.6g   (frames %d, wf_ts dispatches %d)" % (k, wf[k] / frames, frames, n_ts[k]))
err = os.path.join(out, "errors.txt")
if os.path.exists(err):
    print("\n## errors\n" + open(err).read())
