"""Launches of one kernel family in issue order (last proof of a rocprofv3 --kernel-trace CSV) with template arguments.
usage: python tools/launch_seq.py <dir> <kernel substring> <proofs in trace> [max]"""
import csv
import glob
import re
import sys

base, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
mx = int(sys.argv[4]) if len(sys.argv) > 4 else 60
f = glob.glob(f"{base}/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // n
for r in rows[-per:][:mx]:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3:8.1f} us  grid {r.get('Grid_Size_X', '?'):>9s}  {nm[:90]}")
