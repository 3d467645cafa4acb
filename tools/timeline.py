"""Where one proof's wall time goes: kernel time and the idle gap that FOLLOWS each kernel family (host round trips), from a
rocprofv3 --kernel-trace CSV holding `--steps` identical steps of `bench.py --inflight 1` (last step analysed).
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [family-for-per-launch-listing]"""
import csv
import glob
import re
import sys

f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_steps = max(1, sum(1 for r in rows if "k_logup_fill" in r["Kernel_Name"]))
# one WHOLE-NODE step: from the step's VM batch (the last k_vm_segments launch of the trace) to the end of the trace; traces without
# the device VM: from the last stacking copy / commit on
vm = [i for i, r in enumerate(rows) if "k_vm_segments" in r["Kernel_Name"]]
first = [i for i, r in enumerate(rows) if "k_stack_columns" in r["Kernel_Name"]]
step = rows[vm[-1]:] if vm else (rows[first[-1]:] if first else rows[-(len(rows) // n_steps):])


def fam(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return re.sub(r"<.*", "", n)


t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
agg = {}
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, int(step[i + 1]["Start_Timestamp"]) - e) if i + 1 < len(step) else 0
    a = agg.setdefault(fam(r["Kernel_Name"]), [0, 0.0, 0.0, 0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += gap / 1e3
    a[3] += 1 if (e - s) < 20000 else 0
busy = sum(a[1] for a in agg.values())
print(f"steps {n_steps}; last step span {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e3:.3f} ms, idle {(t1 - t0) / 1e6 - busy / 1e3:.3f} ms, "
      f"launches {len(step)}, launches < 20 us: {sum(a[3] for a in agg.values())}")
print(f"{'kernel family':34s} {'n':>5s} {'kernel ms':>10s} {'gap-after ms':>13s} {'<20us':>6s}")
for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{n[:34]:34s} {a[0]:5d} {a[1] / 1e3:10.3f} {a[2] / 1e3:13.3f} {a[3]:6d}")
if len(sys.argv) > 2:
    for i, r in enumerate(step):
        if sys.argv[2] in r["Kernel_Name"]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = int(step[i + 1]["Start_Timestamp"]) - e if i + 1 < len(step) else 0
            nm = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
            print(f"{(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f} us  grid {r.get('Grid_Size_X', '?'):>8s}  {nm[:70]}")
