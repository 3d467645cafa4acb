"""Per-kernel-family VALU work from one rocprofv3 --pmc SQ_INSTS_VALU pass of `bench.py --inflight 1` (one prover alone):
wave-level VALU instructions x 64 lanes against the chip's issue rate 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s
(/opt/skills/guides/MI355X_MICROARCH.md), unweighted and weighted by the issue cycles of each kernel's static ISA mix
(profiles/r02_isa_mix.json from tools/isa_mix.py --all: v_mul_lo/hi_u32 and v_mad_u64_u32 take 4 cycles per wave64,
v_lshl_add_u64 7.4, everything else 2).
usage: python tools/valu_summary.py <dir with pmc_valu/> <proofs in trace> <out.json> [isa_mix.json]"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

base, n_steps, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
mix = json.load(open(sys.argv[4]))["kernels"] if len(sys.argv) > 4 else {}
PEAK = 256 * 4 * 32 * 2.4e9  # lane-instructions / s


def family(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]


def weight(name):
    k = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    return mix[k]["issue_cycle_weight"] if k in mix else None


f = glob.glob(f"{base}/pmc_valu/**/*_counter_collection.csv", recursive=True)[0]
agg = {}
LARGE_NS = 100_000  # "large launches": the throughput-bound part of a family, without its latency-bound tail of tiny launches
large = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU":
        continue
    a = agg.setdefault(family(r["Kernel_Name"]), [0, 0.0, 0, 0.0])
    w = weight(r["Kernel_Name"])
    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
    a[2] += ns
    a[3] += float(r["Counter_Value"]) * (w if w else 1.0)
    if ns >= LARGE_NS:
        b = large.setdefault(family(r["Kernel_Name"]), [0, 0.0, 0, 0.0])
        b[0] += 1
        b[1] += float(r["Counter_Value"])
        b[2] += ns
        b[3] += float(r["Counter_Value"]) * (w if w else 1.0)
out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline --inflight 1",
       "note": "SQ_INSTS_VALU counts wave-level instructions; x64 lanes; peak = 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s; "
               "issue_cycle_weight = issue cycles per instruction / 2 from the static ISA mix of each kernel instantiation",
       "source_sha": bench.source_sha(), "proofs_in_trace": n_steps, "per_proof": {}}
tot_alu, tot_w, tot_ms = 0.0, 0.0, 0.0
for k, (n, v, ns, vw) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    alu_ms = v * 64 / PEAK * 1e3 / n_steps
    w_ms = vw * 64 / PEAK * 1e3 / n_steps
    ms = ns * 1e-6 / n_steps
    tot_alu += alu_ms
    tot_w += w_ms
    tot_ms += ms
    out["per_proof"][k] = {"launches": n / n_steps, "valu_wave_insts": v / n_steps, "issue_cycle_weight": round(vw / v, 3) if v else None,
                           "alu_roofline_ms": round(alu_ms, 4), "alu_roofline_issue_weighted_ms": round(w_ms, 4),
                           "kernel_ms_under_pmc": round(ms, 4), "alu_frac": round(alu_ms / ms, 3) if ms else None,
                           "alu_frac_issue_weighted": round(w_ms / ms, 3) if ms else None}
out["large_launches"] = {"definition": "launches of >= 100 us under the counter pass", "per_proof": {}}
for k, (n, v, ns, vw) in sorted(large.items(), key=lambda kv: -kv[1][2]):
    ms = ns * 1e-6 / n_steps
    out["large_launches"]["per_proof"][k] = {"launches": n / n_steps, "kernel_ms_under_pmc": round(ms, 4),
                                             "alu_frac": round(v * 64 / PEAK * 1e3 / n_steps / ms, 3),
                                             "alu_frac_issue_weighted": round(vw * 64 / PEAK * 1e3 / n_steps / ms, 3)}
out["total_alu_roofline_ms_per_proof"] = round(tot_alu, 3)
out["total_alu_roofline_issue_weighted_ms_per_proof"] = round(tot_w, 3)
out["total_kernel_ms_per_proof_under_pmc"] = round(tot_ms, 3)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("total_alu_roofline_ms_per_proof", "total_alu_roofline_issue_weighted_ms_per_proof",
                                      "total_kernel_ms_per_proof_under_pmc")}))
for k, v in list(out["per_proof"].items())[:16]:
    print(f"{k:26s} alu {v['alu_roofline_ms']:7.3f} ms  weighted {v['alu_roofline_issue_weighted_ms']:7.3f} ms  kernel {v['kernel_ms_under_pmc']:7.3f} ms  "
          f"frac {v['alu_frac']}  weighted {v['alu_frac_issue_weighted']}")
print("launches of >= 100 us only:")
for k, v in out["large_launches"]["per_proof"].items():
    print(f"  {k:26s} {v['launches']:5.1f} launches  kernel {v['kernel_ms_under_pmc']:7.3f} ms  frac {v['alu_frac']}  weighted {v['alu_frac_issue_weighted']}")
