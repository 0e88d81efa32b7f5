#!/usr/bin/env python3
"""Compact view of a rocprofv3 `*_kernel_stats.csv`: our kernels, rocPRIM and runtime fills/copies only.

usage: python tools/kstats.py gpurun_out/prof_x/x_kernel_stats.csv
"""
import csv
import re
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if not any(s in n for s in ("drprg", "rocprim", "rocclr")):
        continue
    short = re.sub(r"\(.*", "", re.sub(r"<.*", "", n.replace("void ", "")))[:58]
    if "rocprim" in n:
        m = re.search(r"detail::(\w+)<", n[60:])
        short = "rocprim:" + (m.group(1) if m else "?")
    print(f"{short:58s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:9.1f} "
          f"min={float(r['MinNs']) / 1e3:8.1f} max={float(r['MaxNs']) / 1e3:8.1f}")
