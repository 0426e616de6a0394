#!/usr/bin/env python3
"""Prints the kernel timeline of the last frame of each wavefront workload in a rocprofv3 --kernel-trace csv.
usage: scripts/wf_timeline.py <dir with *kernel_trace.csv> [feat ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for feat in (sys.argv[2:] or ["0", "1"]):
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void wf_ts<false, %s>" % feat)]
    if not idx:
        continue
    # frames are separated by a wf_gather run; take the last complete frame
    last = idx[-1]
    start = last
    while start > 0 and not rows[start - 1]["Kernel_Name"].startswith("wf_gather"):
        start -= 1
    end = last
    while end + 1 < len(rows) and (rows[end + 1]["Kernel_Name"].startswith("void wf_") or rows[end + 1]["Kernel_Name"].startswith("wf_gather")):
        end += 1
    t0 = int(rows[start]["Start_Timestamp"])
    print("== feature level %s: %d kernels, %.1f us" % (feat, end - start + 1, (int(rows[end]["End_Timestamp"]) - t0) / 1e3))
    for r in rows[start:end + 1]:
        print("  %-22s start %8.1f us  dur %7.1f us" % (r["Kernel_Name"].replace("void ", "")[:22], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
