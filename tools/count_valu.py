"""Static VALU instruction count per (row pair, evaluation point) of every k_air_round instantiation, from the gfx950 ISA.
The kernels are straight-line inside the grid-stride loop (the Poseidon segments are fully unrolled), so the static count
of v_* instructions is the per-evaluation count up to the loop prologue/epilogue (< 1 %).
usage: python tools/count_valu.py > profiles/r01_air_valu_counts.json"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "leanmultisig_amd", "csrc", "lm_air.hip")
asm = "/tmp/lm_air_count.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", asm],
                      stderr=subprocess.DEVNULL)
out, cur, has_loop = {}, None, {}
for line in open(asm):
    m = re.match(r"^(_Z11k_air_roundILi(\d)E(j8BaseCols|N2kb2EFE7ExtCols)Li(n?\d)E\S*):", line)
    if m:
        cur = f"table{m.group(2)}_{'base' if m.group(3).startswith('j') else 'ef'}_seg{m.group(4).replace('n', '-')}"
        out[cur] = 0
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        cur = None
    elif t.startswith("v_"):
        out[cur] += 1
res = {}
for t in (0, 1, 2):
    for f in ("base", "ef"):
        if t == 2:
            res[f"table{t}_{f}"] = sum(out[f"table2_{f}_seg{s}"] for s in range(5))
        else:
            res[f"table{t}_{f}"] = out[f"table{t}_{f}_seg-1"]
json.dump({"valu_per_evaluation": res, "per_kernel": out,
           "note": "v_* instructions in the ISA of each k_air_round<TABLE, T, Cols, SEG> (hipcc -O3, gfx950); table 2 = sum of its 5 segments"},
          sys.stdout, indent=1)
