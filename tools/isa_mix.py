"""Static VALU instruction mix per kernel from hipcc's gfx950 assembly (`hipcc -S --cuda-device-only`).
Issue-cycle weight = sum(count x cycles per wave64) / (2 x count): v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 issue in 4
cycles, v_lshl_add_u64 at its measured 7.4 (tools/ubench/int_rates.hip, profiles/r02_int_rates.txt), everything else in 2
(SIMD-32, /opt/skills/guides/MI355X_MICROARCH.md).  Straight-line kernels: static mix ~ dynamic mix.
usage: python tools/isa_mix.py file.s [name-substring]"""
import collections
import json
import re
import sys

CYCLES = {"v_mul_lo_u32": 4, "v_mul_hi_u32": 4, "v_mad_u64_u32": 4, "v_mul_hi_i32": 4, "v_mad_i64_i32": 4, "v_lshl_add_u64": 7.4}


def mixes(path, want=""):
    txt = open(path).read()
    parts = re.split(r"\n(_Z[0-9A-Za-z_]+):[^\n]*\n", txt)
    out = {}
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
        if want not in name:
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            m = re.match(r"\s+(v_\w+|s_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", line)
            if m:
                c[m.group(1)] += 1
        valu = {k: v for k, v in c.items() if k.startswith("v_")}
        n = sum(valu.values())
        if not n:
            continue
        cyc = sum(v * CYCLES.get(k, 2) for k, v in valu.items())
        out[name] = dict(valu=n, mul_class=sum(v for k, v in valu.items() if k in CYCLES and k != "v_lshl_add_u64"),
                         lshl_add_u64=valu.get("v_lshl_add_u64", 0), dpp=sum(v for k, v in valu.items() if "dpp" in k),
                         readlane=valu.get("v_readlane_b32", 0) + valu.get("v_writelane_b32", 0), s_nop=c.get("s_nop", 0),
                         issue_cycle_weight=round(cyc / (2 * n), 3), top=c.most_common(8))
    return out


if __name__ == "__main__":
    for k, v in mixes(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "").items():
        print(k[:60], json.dumps(v))
