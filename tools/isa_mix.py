"""Static VALU instruction mix per kernel from hipcc's gfx950 assembly (`hipcc -S --cuda-device-only`).
Issue-cycle weight = sum(count x cycles per wave64) / (2 x count): v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 issue in 4
cycles, v_lshl_add_u64 at its measured 7.4 (tools/ubench/int_rates.hip, profiles/r02_int_rates.txt), everything else in 2
(SIMD-32, /opt/skills/guides/MI355X_MICROARCH.md).  Straight-line kernels: static mix ~ dynamic mix.
usage: python tools/isa_mix.py file.s [name-substring]
       python tools/isa_mix.py --all out.json     (compiles every csrc/*.hip to assembly; keys = demangled kernel names)"""
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


def all_kernels(out_path):
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    res = {}
    for f in sorted(os.listdir(os.path.join(root, "leanmultisig_amd", "csrc"))):
        if not f.endswith(".hip"):
            continue
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                   os.path.join(root, "leanmultisig_amd", "csrc", f), "-o", tmp.name], stderr=subprocess.DEVNULL)
            mx = mixes(tmp.name)
        names = list(mx)
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        for n, d in zip(names, dem):
            v = mx[n]
            v.pop("top")
            res[d.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]] = v
    # one permutation per lane, straight line (tools/ubench/one_perm.hip): the issue-bound ceiling of the leaf sponge is derived from THIS
    # count (the sponge kernel itself is a loop around it: its static count is not per permutation)
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               os.path.join(root, "tools", "ubench", "one_perm.hip"), "-o", tmp.name], stderr=subprocess.DEVNULL)
        v = next(iter(mixes(tmp.name, "k_one_perm").values()))
        v.pop("top")
        res["poseidon16_permute_one_lane"] = v
    json.dump({"source_sha": bench.source_sha(), "cycles_per_wave64": {**CYCLES, "other": 2}, "kernels": res}, open(out_path, "w"), indent=1)
    print("wrote", out_path, len(res), "kernels")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--all":
        all_kernels(sys.argv[2])
        sys.exit(0)
    for k, v in mixes(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "").items():
        print(k[:60], json.dumps(v))
