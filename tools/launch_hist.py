"""Per-launch durations of one kernel family from a rocprofv3 --kernel-trace CSV: the longest launches and the tail.
usage: python tools/launch_hist.py <dir> <kernel substring> <proofs in trace>"""
import csv
import glob
import sys

base, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(f"{base}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if pat == "":  # every launch of ONE whole-node step: from the last k_vm_segments launch of the trace (the step's VM batch) to its end
    start = max(i for i, r in enumerate(rows) if "k_vm_segments" in r["Kernel_Name"])
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows[start:]]
    per = len(d)
else:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows if pat in r["Kernel_Name"]]
    per = len(d) // n
last = sorted(d[-per:], reverse=True)
print(f"{pat or 'ALL KERNELS'}: {per} launches per proof, {sum(last) * 1e-3:.3f} ms")
print("longest (us):", " ".join(f"{x:.0f}" for x in last[:40]))
for lim in (5, 10, 20, 50):
    sel = [x for x in last if x < lim]
    print(f"  < {lim:3d} us: {len(sel):4d} launches, {sum(sel) * 1e-3:.3f} ms")
