"""Per-kernel-family VALU work from one rocprofv3 --pmc SQ_INSTS_VALU pass of the bench command (one prover alone):
wave-level VALU instructions x 64 lanes / peak issue rate = the time the family would take at the integer-ALU roofline.
usage: python tools/valu_summary.py <dir with pmc_valu/> <proofs in trace> <out.json>"""
import csv
import glob
import json
import sys

base, n_steps, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
PEAK = 256 * 4 * 16 * 2.4e9  # lane-instructions / s


def family(name):
    return name.replace("void ", "").split("(")[0].split("<")[0]


f = glob.glob(f"{base}/pmc_valu/*/*_counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU":
        continue
    a = agg.setdefault(family(r["Kernel_Name"]), [0, 0.0, 0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
    a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline --inflight 1",
       "note": "SQ_INSTS_VALU counts wave-level instructions; x64 lanes; peak = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz",
       "proofs_in_trace": n_steps, "per_proof": {}}
tot_alu, tot_ms = 0.0, 0.0
for k, (n, v, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    alu_ms = v * 64 / PEAK * 1e3 / n_steps
    ms = ns * 1e-6 / n_steps
    tot_alu += alu_ms
    tot_ms += ms
    out["per_proof"][k] = {"launches": n / n_steps, "valu_wave_insts": v / n_steps, "alu_roofline_ms": round(alu_ms, 4),
                           "kernel_ms_under_pmc": round(ms, 4), "alu_frac": round(alu_ms / ms, 3) if ms else None}
out["total_alu_roofline_ms_per_proof"] = round(tot_alu, 3)
out["total_kernel_ms_per_proof_under_pmc"] = round(tot_ms, 3)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("total_alu_roofline_ms_per_proof", "total_kernel_ms_per_proof_under_pmc")}))
for k, v in list(out["per_proof"].items())[:14]:
    print(f"{k:28s} alu {v['alu_roofline_ms']:8.3f} ms  kernel {v['kernel_ms_under_pmc']:8.3f} ms  frac {v['alu_frac']}")
