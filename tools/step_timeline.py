"""prints the kernel timeline of the last bench step out of a rocprofv3 --kernel-trace csv: start (us from the step's first
kernel), duration, gap to the previous kernel's end -- where the time of a step goes that no kernel accounts for"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first_name = sys.argv[2] if len(sys.argv) > 2 else "sketch_filter_kernel"
starts = [i for i, r in enumerate(rows) if first_name in r["Kernel_Name"]]
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for si in starts[-n_steps:]:
    # the step's launches: from the memsets just before its first kernel to the next step's start
    lo = si
    while lo > 0 and int(rows[si]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 60000 and first_name not in rows[lo - 1]["Kernel_Name"]:
        lo -= 1
    nxt = [s for s in starts if s > si]
    hi = nxt[0] if nxt else len(rows)
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    print("---- step")
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s - t0 > 3_000_000:
            break
        name = r["Kernel_Name"].split("(")[0].split("<")[0][-44:]
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
        prev_end = max(prev_end, e)
    print(f"   span {(prev_end - t0) / 1e3:.1f} us")
