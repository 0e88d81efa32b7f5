#!/usr/bin/env python3
"""profiles/traffic.json and profiles/valu.json from the trimmed rocprofv3 counter files of one profile run.

usage: python tools/make_profile_json.py profiles/r05 z          (files z_pmc_<workload>_{FETCH,WRITE}_SIZE_counter_collection.csv,
                                                                  z_sq_<workload>_p{1,2}_counter_collection.csv)
Entries of workloads the run did not measure are kept from the existing files.  Every entry carries the sha256 (16 hex digits) of the
kernel's source files at the moment this script runs -- run it at the commit the profiles were taken on; bench.py drops an entry whose
sources have changed since (roofline.traffic / roofline.secondary = null, with the reason)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = {  # (bench.py holds the same table)
    "sketch_filter_kernel": ["sketch_filter.hip", "filter_common.h", "device_common.h", "kernels.h", "common.h"],
    "sketch_wave_kernel": ["sketch_wave.hip", "sketch_block.h", "device_common.h", "kernels.h", "common.h"],
    "sketch_probe_kernel": ["sketch_probe.hip", "device_common.h", "kernels.h", "common.h"],
}
DOMINANT = {"big": "sketch_wave_kernel", "mtb-x16": "sketch_wave_kernel", "mtb-x32": "sketch_wave_kernel"}


def source_sha16(kernel):
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[kernel]:
        h.update(open(os.path.join(ROOT, "drprg_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def per_launch(path, kernel):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if kernel + "<" in r["Kernel_Name"] or r["Kernel_Name"].endswith(kernel) or (kernel + "(") in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: v / n for k, (v, n) in acc.items()}


def main():
    d, tag = sys.argv[1], sys.argv[2]
    tfile, vfile = os.path.join(ROOT, "profiles", "traffic.json"), os.path.join(ROOT, "profiles", "valu.json")
    traffic, valu = json.load(open(tfile)), json.load(open(vfile))
    rel = os.path.relpath(d, ROOT)
    for wl in sorted({re.match(rf"{tag}_pmc_(.+)_(FETCH|WRITE)_SIZE", os.path.basename(f)).group(1) for f in glob.glob(f"{d}/{tag}_pmc_*_SIZE_counter_collection.csv")}):
        k = DOMINANT.get(wl, "sketch_filter_kernel")
        fe = per_launch(f"{d}/{tag}_pmc_{wl}_FETCH_SIZE_counter_collection.csv", k).get("FETCH_SIZE")
        wr = per_launch(f"{d}/{tag}_pmc_{wl}_WRITE_SIZE_counter_collection.csv", k).get("WRITE_SIZE")
        if fe is None or wr is None:
            continue
        doubled = k != "sketch_wave_kernel"  # (random 4/16-byte probes: undoubled, see _comment)
        traffic[wl] = {k: {"FETCH_SIZE_KiB": round(fe, 1), "WRITE_SIZE_KiB": round(wr, 1), "hbm_bytes_per_launch": int((fe * (2 if doubled else 1) + wr) * 1024),
                           "source_sha16": source_sha16(k), "rows": f"{rel}/{tag}_pmc_{wl}_*"}}
    for wl in sorted({re.match(rf"{tag}_sq_(.+)_p\d", os.path.basename(f)).group(1) for f in glob.glob(f"{d}/{tag}_sq_*_counter_collection.csv")}):
        k = DOMINANT.get(wl, "sketch_filter_kernel")
        c = {}
        for f in glob.glob(f"{d}/{tag}_sq_{wl}_p*_counter_collection.csv"):
            c.update(per_launch(f, k))
        if "SQ_INSTS_VALU" not in c:
            continue
        valu[wl] = {k: {"valu_wave_insts_per_launch": c["SQ_INSTS_VALU"], "salu": c.get("SQ_INSTS_SALU"), "lds": c.get("SQ_INSTS_LDS"), "waves": c.get("SQ_WAVES"),
                        "busy_cycles": c.get("SQ_BUSY_CYCLES"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY"), "source_sha16": source_sha16(k),
                        "rows": f"{rel}/{tag}_sq_{wl}_*"}}
    json.dump(traffic, open(tfile, "w"), indent=1)
    json.dump(valu, open(vfile, "w"), indent=1)
    print("traffic:", {w: list(v) for w, v in traffic.items() if not w.startswith("_")})
    print("valu:", {w: list(v) for w, v in valu.items() if not w.startswith("_")})


if __name__ == "__main__":
    main()
