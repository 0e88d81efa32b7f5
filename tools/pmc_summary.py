#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc counter_collection.csv files for our kernels.

usage: python tools/pmc_summary.py gpurun_out/pmc_<tag>
"""
import collections
import csv
import glob
import re
import sys

for f in sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "drprg" not in k and "rocprim" not in k:
            continue
        short = re.sub(r"\(.*", "", re.sub(r"<.*", "", k.replace("void ", "")))[-44:]
        a = acc[(short, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(f"{k:46s} {c:24s} per-launch {v / n:16.1f}  (n={n})")
