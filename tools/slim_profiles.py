#!/usr/bin/env python3
"""Make the rocprofv3 csv files of a profiles/ directory small enough to commit: per-dispatch traces keep kernel name (without its
argument list), start and end; counter files keep the rows of this project's kernels and the columns the summaries read.
usage: python tools/slim_profiles.py profiles/r06"""
import csv
import glob
import re
import sys

d = sys.argv[1]


def short(n):
    return re.sub(r"\(.*", "", n)


for f in glob.glob(f"{d}/*_kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(f, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for r in rows:
            n = r.get("Kernel_Name") or r.get("Name") or ""
            if "drprg" in n or "rocprim" in n:
                w.writerow([short(n), r["Start_Timestamp"], r["End_Timestamp"]])
for f in glob.glob(f"{d}/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(f, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for r in rows:
            if "drprg" in r["Kernel_Name"]:
                w.writerow([short(r["Kernel_Name"]), r["Counter_Name"], r["Counter_Value"]])
