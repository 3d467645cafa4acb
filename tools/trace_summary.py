"""Per-launch durations of selected kernels from a rocprofv3 --kernel-trace CSV (last bench step only).
usage: python tools/trace_summary.py <kernel_trace.csv> [name ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed step starts at the last k_ntt_pass burst preceded by the stacking copies: take the last occurrence of the
# first kernel name of the step (the trace holds warmup + steps, identical sequences)
names = [r["Kernel_Name"] for r in rows]
n_steps = max(1, sum(1 for n in names if "k_logup_fill" in n))
per = len(rows) // n_steps
step = rows[-per:]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"steps in trace {n_steps}; last step: span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, launches {len(step)}")
want = sys.argv[2:]
agg = {}
for r in step:
    n = r["Kernel_Name"].split("(")[0]
    n = n.replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(n, []).append(d)
for n, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if want and not any(w in n for w in want):
        continue
    line = f"{n[:70]:70s} n={len(d):4d} total={sum(d) / 1e3:7.3f} ms"
    if want:
        line += " us: " + " ".join(f"{x:.0f}" for x in d)
    print(line)
