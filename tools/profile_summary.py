#!/usr/bin/env python3
"""summary.txt of a profile directory, written from the csv / json files that sit in it (tools/run_profiles_r06.sh calls it last; run it
again on profiles/r06 after copying: `python tools/profile_summary.py profiles/r06 z_`).  Every kernel line names the file it comes from
and its call count, so that a summary cannot outlive the table it summarises.

usage: python tools/profile_summary.py <dir> [file prefix ...]      (profiles/r06: z_ zb_)"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kstats  # noqa: E402

d = sys.argv[1]
pres = sys.argv[2:] or [""]


def files(pattern):
    return sorted(f for pre in pres for f in glob.glob(f"{d}/{pre}{pattern}"))


print("== bench lines")
for f in files("bench_*.json"):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        fs = r.get("filter_schedule") or {}
        sec = (r.get("secondary") or {}).get("frac")
        print("%-46s ms/step %.4f value %.3e kernel %.4f ms frac %.3f secondary %s traffic %s | %s rounds %s class ends %s | steps %d warmup %d groups %s" % (
            os.path.basename(f), j["ms_per_step"], j["value"], r["avg_launch_ms"], r["frac"], "%.3f" % sec if sec else None, r.get("traffic"),
            fs.get("form"), fs.get("chunk_tiles_per_round"), fs.get("class_end_us"), j["steps"], j["warmup"], j["step_ms"].get("groups")))
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), "FAILED", e)
print("\n== kernels (rocprofv3 --kernel-trace; full = launches >= a tenth of the kernel's median)")
for f in files("*_kernel_trace.csv") or files("*_kernel_stats.csv"):
    print("--", os.path.basename(f))
    for r in kstats.load(f):
        if "rocclr" in r["name"]:
            continue
        line = f"   {r['name']:56s} calls={r['calls']:>4d} avg_us={r['avg_us']:9.1f} min={r['min_us']:8.1f} max={r['max_us']:8.1f}"
        if "full_avg_us" in r:
            line += f" full={r['full_avg_us']:9.1f} (n={r['full_calls']})"
        print(line)
    st = f.replace("_kernel_trace.csv", "_kernel_stats.csv")
    if st != f and os.path.exists(st):  # the two files of one run must agree on the call counts
        a = {r["name"]: r["calls"] for r in kstats.from_trace(f)}
        b = {r["name"]: r["calls"] for r in kstats.from_stats(st)}
        bad = {k: (a.get(k), b.get(k)) for k in set(a) | set(b) if a.get(k) != b.get(k) and "rocprim" not in k and "rocclr" not in k}  # (the runtime's fills and copies are trimmed out of the traces: slim_profiles.py)
        print("   call counts equal those of", os.path.basename(st) + ":", not bad, bad or "")
print("\n== counters (per launch)")
for f in files("*counter_collection.csv"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "drprg" not in k:
            continue
        a = acc[(kstats.short_name(k), r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(f"   {os.path.basename(f)[:44]:44s} {k[-36:]:36s} {c:22s} {v / n:16.1f} (n={n})")
