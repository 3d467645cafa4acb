"""Every launch of the last whole-node step of a rocprofv3 --kernel-trace CSV in start order: offset, duration, gap to the next start,
queue, grid, kernel.  usage: python tools/launch_list.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import re
import sys

f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
vm = [i for i, r in enumerate(rows) if "k_vm_segments" in r["Kernel_Name"]]
step = rows[vm[-1]:] if vm else rows
t0 = int(step[0]["Start_Timestamp"])
end_max = t0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle = max(0, s - end_max)  # time with NO kernel running before this start
    end_max = max(end_max, e)
    nm = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} idle {idle / 1e3:6.1f} q{r.get('Queue_Id', '?'):>2s} grid {r.get('Grid_Size_X', '?'):>9s} wg {r.get('Workgroup_Size_X', '?'):>4s} {nm[:100]}")
