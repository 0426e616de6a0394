#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --pmc output directory: for every kernel name (template arguments kept, argument list cut)
and counter, dispatch count, sum and mean per dispatch.  usage: pmc_by_kernel.py <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    for out in sys.argv[1:]:
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row.get("Kernel_Name", ""))
                c = acc[k][row["Counter_Name"]]
                c[0] += 1
                c[1] += float(row["Counter_Value"])
        print("# %s" % out)
        for k in sorted(acc):
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            for cn in sorted(acc[k]):
                n, s = acc[k][cn]
                print("%-44s %-30s n=%-5d sum=%-14.6g mean=%.6g" % (k[:44], cn, n, s, s / n))


if __name__ == "__main__":
    main()
