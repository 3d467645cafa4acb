"""Per-kernel-family HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X
guide prescribes) of the bench command.  usage: python tools/pmc_summary.py <dir with pmc_fetch/ pmc_write/> <steps+warmup> <out.json>"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

base, n_steps, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def family(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    return n.split("<")[0]


def collect(sub, counter):
    f = glob.glob(f"{base}/{sub}/**/*_counter_collection.csv", recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = family(r["Kernel_Name"])
        a = agg.setdefault(k, [0, 0.0, 0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


fetch, write = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
out = {
    "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --inflight 1 (two separate passes)",
    "unit_note": "counters are KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes read, for EVERY access width this library "
                 "uses: calibrated with tools/ubench/fetch_calib.hip on 1 GiB streams (profiles/r04_fetch_calibration.txt: 4, 8 and 16 bytes "
                 "per lane 0.500, the 5-plane uint2 pattern of k_air_round / k_gkr_step 0.526 = 0.500 + the over-fetch of its plane tails) — "
                 "so x2 is applied to all kernels.  WRITE_SIZE matches byte counts exactly (1.000 for 4 and 16 bytes per lane).",
    "source_sha": bench.source_sha(),
    "steps_in_trace": n_steps,
    "per_step": {},
}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0, 0]), write.get(k, [0, 0.0, 0])
    launches = max(f[0], w[0]) / n_steps
    out["per_step"][k] = {
        "launches": launches,
        "fetch_KiB_raw": f[1] / n_steps,
        "fetch_bytes_x2": 2 * 1024 * f[1] / n_steps,
        "write_bytes": 1024 * w[1] / n_steps,
        "hbm_bytes": (2 * f[1] + w[1]) * 1024 / n_steps,
        "kernel_ms_under_pmc": f[2] / n_steps / 1e6,
    }
json.dump(out, open(out_path, "w"), indent=1)
top = sorted(out["per_step"].items(), key=lambda kv: -kv[1]["hbm_bytes"])[:14]
for k, v in top:
    print(f"{k:28s} launches/step {v['launches']:7.1f}  HBM {v['hbm_bytes'] / 1e6:9.1f} MB/step  ({v['kernel_ms_under_pmc']:.2f} ms)")
