#!/usr/bin/env python3
"""counter value / known bytes for tools/ubench/fetch_calib.hip.  usage: python tools/fetch_calib_summary.py <dir with out_f/ out_w/>"""
import csv
import glob
import sys

base = sys.argv[1]
WORDS = 1 << 28
known = {"read4": WORDS * 4, "read8": WORDS * 4, "read16": WORDS * 4, "read8_planes5": WORDS // 5 // 2 * 2 * 5 * 4, "write4": WORDS * 4, "write16": WORDS * 4}
print("rocprofv3 counters against known byte counts (1 GiB buffers, gfx950); counter unit = KiB")
for sub, counter in (("out_f", "FETCH_SIZE"), ("out_w", "WRITE_SIZE")):
    f = glob.glob(f"{base}/{sub}/**/*_counter_collection.csv", recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        a = agg.setdefault(k, [0, 0.0, 0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in sorted(agg):
        if k not in known:
            continue
        n, v, ns = agg[k]
        per = v / n * 1024
        moved = known[k]
        print(f"{counter:11s} {k:14s} launches {n}  counter {per / 1e6:10.1f} MB  known {moved / 1e6:10.1f} MB  counter/known {per / moved:6.3f}  "
              f"({moved / (ns / n) :7.1f} GB/s under the counter pass)")
