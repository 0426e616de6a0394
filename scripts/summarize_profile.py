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
print("\n## PMC passes (per-dispatch mean over rtc_trace_kernel<false> dispatches; counting variant <true> excluded)")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        acc, n = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "rtc_trace_kernel<false" not in kn and "rtc_persist_kernel<false" not in kn:
                continue
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
        for k in sorted(acc):
            print("%-32s mean/dispatch = %.6g   (dispatches %d)" % (k, acc[k] / n[k], n[k]))
err = os.path.join(out, "errors.txt")
if os.path.exists(err):
    print("\n## errors\n" + open(err).read())
