#!/usr/bin/env python3
"""Compact view of a rocprofv3 `*_kernel_stats.csv` -- or, better, of the `*_kernel_trace.csv` next to it: our kernels, rocPRIM and
runtime fills/copies only.

usage: python tools/kstats.py <x_kernel_stats.csv | x_kernel_trace.csv> [--json]

From a trace (one row per dispatch) the line also carries `full=`: the average over the launches that did a batch's work, i.e. without
the launches shorter than a tenth of the kernel's median (read_cluster_kernel runs once or twice per bench with next to nothing -- the
checks' small batches, an empty last chunk: ~5 us launches that pulled the plain average of the round-5 tables 9 % low).
Given a stats file, the trace of the same prefix is used when it exists.
"""
import csv
import json
import os
import re
import statistics
import sys


def short_name(n):
    short = re.sub(r"\(.*", "", re.sub(r"<.*", "", n.replace("void ", "")))[:58]
    if "rocprim" in n:
        m = re.search(r"detail::(\w+)<", n[60:])
        short = "rocprim:" + (m.group(1) if m else "?")
    return short


def ours(n):
    return any(s in n for s in ("drprg", "rocprim", "rocclr"))


def from_trace(path):
    per = {}
    for r in csv.DictReader(open(path)):
        n = r.get("Kernel_Name") or r.get("Name") or ""
        if not ours(n):
            continue
        per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for n, d in per.items():
        med = statistics.median(d)
        full = [x for x in d if x >= 0.1 * med]
        rows.append({"name": short_name(n), "calls": len(d), "avg_us": sum(d) / len(d), "min_us": min(d), "max_us": max(d), "median_us": med,
                     "full_calls": len(full), "full_avg_us": sum(full) / len(full), "total_us": sum(d)})
    return sorted(rows, key=lambda r: -r["total_us"])


def from_stats(path):
    rows = []
    for r in csv.DictReader(open(path)):
        if ours(r["Name"]):
            rows.append({"name": short_name(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                         "max_us": float(r["MaxNs"]) / 1e3})
    return rows


def load(path):
    trace = path if path.endswith("_kernel_trace.csv") else path.replace("_kernel_stats.csv", "_kernel_trace.csv")
    return from_trace(trace) if os.path.exists(trace) else from_stats(path)


if __name__ == "__main__":
    rows = load(sys.argv[1])
    if "--json" in sys.argv:
        print(json.dumps(rows))
    else:
        for r in rows:
            line = f"{r['name']:58s} calls={r['calls']:>4d} avg_us={r['avg_us']:9.1f} min={r['min_us']:8.1f} max={r['max_us']:8.1f}"
            if "full_avg_us" in r:
                line += f" full={r['full_avg_us']:9.1f} (n={r['full_calls']}) median={r['median_us']:9.1f}"
            print(line)
