#!/usr/bin/env python3
"""bench.py — XMSS signatures aggregated per second, whole node, 1550 signatures, WHIR rate 1/2 (BASELINE configs[1]).

Default workload: the hand-assembled XMSS aggregation program (leanmultisig_amd/programs/xmss_aggregate.py: the reference's
zkDSL program at the ISA level, looped, 2^19-row bytecode table) on 1550 REAL signatures.
  value             one `lmh_prove_execution_vm` per step = what the reference's metric times (aggregate_type_1 ->
                    prove_execution(bytecode, public_input, witness): VM run + trace generation + proof,
                    rec_aggregation/src/benchmark.rs:397-431), from hints in host memory to the pruned proof: the leanVM runner
                    (sequential parts on the host, the 1549 per-signature segments of the parallel batch one wavefront each on the
                    device), every table column built on the device from the resident log, stack + WHIR commit (LDE 2^20 x 128,
                    Merkle), logup fill + GKR over 2^25 pairs + column evaluations, batched AIR sumcheck (3 tables), 252-claim WHIR
                    open with 124-bit parameters.  The same definition at every N (one leaf per rank);
  hot_path.value    one `lmh_prove_execution` per step: from an execution trace resident in HBM to the proof (rounds 1-3's headline).
Not built: the zkDSL compiler (config["missing"]).

Multi-GPU (north_star / SURVEY.md §8(e)): independent 1550-signature leaves, one per GPU, no data-path collective; the
only exchange is an RCCL all-gather of the commitment roots and the pruned proofs at the end of each step.  scaling = "weak".

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SIGS = 1550
P = 0x7F000001
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6300.0  # ... of which a wide streaming copy reaches ~6.3 TB/s


def signer_ranges(n_total, world):
    """Partition of the sorted signer set into `world` contiguous leaves."""
    base, rem = divmod(n_total, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


# BASELINE configs[3] stand-in (`recursion --n 4 --log-inv-rate 2`): the root's program is the in-VM verifier of four child proofs,
# which cannot be compiled here (no zkDSL compiler); its table shapes and its mix of ExtensionOp calls are DERIVED by counting
# (tools/recursion_shape.py: Merkle openings, leaf folds, eq factors, sumcheck / GKR / AIR verification per child from the library's own
# WhirConfig::new and the loop structure of the zkDSL sources): execution 2^19, ExtensionOp 2^18, Poseidon16 2^16, memory 2^22.
def recursion_shape():
    from tools import recursion_shape as rs
    return rs.derive()


def build_vm_workload(ctx, rng, n_sigs, log_inv_rate, capacity, log_bytecode=19):
    """The default workload: real signatures -> hints -> lmh_execute_bytecode -> lmh_get_execution_trace (resident in HBM)."""
    import leanmultisig_amd as lm
    from leanmultisig_amd import vm
    from leanmultisig_amd.programs import xmss_aggregate as xa
    bc = xa.build_program(log_bytecode)
    signer = xa.Xmss(compress=lambda x: ctx.poseidon16(x, compress=True))  # key generation / signing hashes on the device (untimed)
    pi, wit, info = xa.build_witness(bc, n_sigs, rng, xmss=signer)
    ex = vm.execute(bc, pi, wit, ctx=ctx)
    dt = vm.DeviceTrace(ctx, bc, ex, pi, log_inv_rate)
    tr = dt.view
    n_vars = ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))
    lm_builder = lm.WhirBuilder.default(log_inv_rate, prox_gaps_conjecture=capacity)
    cfg = lm.WhirConfig.new(lm_builder, n_vars)
    # aggregate_type_1 takes the (public key, signature) pairs in ANY order (it sorts them, type_1_aggregation.rs:232): the step gets them shuffled
    raw = vm.pack_xmss_signatures(info["sig"])[np.random.default_rng(5).permutation(n_sigs)]
    w = dict(n_sigs=n_sigs, log_rows={t: int(tr.tables[t].log_rows) for t in range(3)}, log_memory=int(tr.log_memory), memory_words=int(ex.memory_len), log_bytecode=bc.log_size,
             ending_pc=bc.ending_pc, public_input=pi, bytecode_hash=bc.hash(), bytecode=bc.multilinear,
             counts=dict(poseidon=ex.n_poseidon_calls, extension_op=ex.n_extension_rows, cycles=ex.n_cycles, **ex.counts))
    return dict(w=w, tr=tr, keep=[dt, ex], cfg=cfg, cfgd=cfg.to_dict(), n_vars=n_vars, lm_builder=lm_builder, vm=dict(bc=bc, pi=pi, wit=wit, info=info, raw=raw, message=info["message"], slot=info["slot"]),
                log_inv_rate=log_inv_rate, capacity=capacity)


def build_workload(ctx, orc, ob, rng, scale_log=0, log_inv_rate=1, shape="xmss", capacity=False, witness="xmss"):
    """Side workloads (--witness synthetic / --shape recursion): consistent synthetic leanVM traces written by tests/synth_witness.py
    (straight-line programs of precompile calls + the VM's own padding rows), uploaded once.  shape = recursion: the BASELINE
    configs[3] stand-in of SURVEY.md §8(d)."""
    import leanmultisig_amd as lm
    from tests import synth_witness
    sh = scale_log  # --scale-log k shrinks every table by 2^k (smoke / CI)
    n_calls = (N_SIGS * 167) >> sh

    def fill_rows(rows):  # Poseidon trace rows on the device (lm_poseidon_trace), columns 25.. are overwritten
        cols = [ctx.to_device(np.ascontiguousarray(rows[:, c])) for c in range(109)]
        ctx.poseidon_trace(cols, rows.shape[0])
        for c in range(25, 109):
            rows[:, c] = cols[c].download()

    if shape == "recursion":
        d = recursion_shape()
        sp, root = d["shape"], d["root"]
        ext = [(op, be, size, max(1, cnt >> sh)) for op, be, size, cnt in d["ext_calls"]]
        n_pos_calls, n_ext_calls = root["poseidon_calls"] >> sh, sum(c for _, _, _, c in ext)
        # (the synthetic generator gives every ExtensionOp call fresh operands: its memory needs one more doubling than the executed
        # program's 2^21 words — the stand-in runs at 2^22, as it did before the derived memory size was corrected)
        w = synth_witness.build(orc, rng, n_calls=n_pos_calls, n_blocks=4096 >> min(sh, 6), log_exec=sp["log_exec"] - sh, log_pos=sp["log_pos"] - sh,
                                log_ext=sp["log_ext"] - sh, log_memory=max(sp["log_memory"] + 1 - sh, 16), log_bytecode=sp["log_bytecode"] - sh,
                                fill_rows=fill_rows, n_arith=max(0, (root["cycles"] >> sh) - n_pos_calls - n_ext_calls), ext_calls=ext)
    else:
        w = synth_witness.build(orc, rng, n_calls=n_calls, n_blocks=4096 >> min(sh, 6), log_exec=20 - sh, log_pos=18 - sh,
                                log_ext=8, log_memory=max(20 - sh, 16), log_bytecode=19 - sh, fill_rows=fill_rows)
    w["log_inv_rate"] = log_inv_rate
    tr, keep = lm.make_execution_trace(ctx, w)
    n_vars = ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))
    # default_whir_config: 124-bit, 16 PoW bits, fold 7/5 (lean_prover/src/lib.rs:22-50), integers from the library's own
    # WhirConfig::new; `builder` is the same parameter set in the oracle's format, for the checker (--verify)
    lm_builder = lm.WhirBuilder.default(log_inv_rate, prox_gaps_conjecture=capacity)
    cfg = lm.WhirConfig.new(lm_builder, n_vars)
    return dict(w=w, tr=tr, keep=keep, cfg=cfg, cfgd=cfg.to_dict(), n_vars=n_vars, lm_builder=lm_builder, log_inv_rate=log_inv_rate, capacity=capacity)


def whir_recursion_bench(ctx, lm, args, ob=None, orc=None):
    """--shape whir-recursion: BASELINE configs[3] (`recursion --n 4 --log-inv-rate 2`, src/main.rs:91-115) as far as the recursion program
    is assembled (leanmultisig_amd/programs/whir_verify.py: the PCS opening of the in-VM verifier, zkdsl_implem/whir.py).  Four leaves of
    775 REAL signatures are proved by this library at rate 1/4 (untimed here; their time is reported), their raw transcripts, opening
    claims and un-pruned Merkle openings become the hints of the root step, and ONE step = prove_execution of the root program: the VM
    run (every (child, query) loop a batch of wavefronts on the device), the trace, the proof.  Prints one JSON line (a side measurement:
    root proofs per second; NOT the headline metric) with the measured cycle / Poseidon / ExtensionOp counts of the run next to the
    counts tools/recursion_shape.py derived for the same terms."""
    from leanmultisig_amd import capi, vm
    from leanmultisig_amd.programs import whir_verify as wv
    from leanmultisig_amd.programs import xmss_aggregate as xa
    n_children, child_sigs, rate = 4, max(8, 775 >> args.scale_log), args.log_inv_rate
    capacity = args.soundness == "capacity"
    leaf = xa.build_program(19 if args.scale_log == 0 else None)
    builder = lm.WhirBuilder.default(rate, prox_gaps_conjecture=capacity)
    inst = dict(log_bytecode=leaf.log_size, ending_pc=leaf.ending_pc, bytecode_hash=leaf.hash(), bytecode=leaf.multilinear)
    signer = xa.Xmss(compress=lambda x: ctx.poseidon16(x, compress=True))
    children, leaf_ms, n_vars = [], [], None
    for c in range(n_children):
        pi, wit, _ = xa.build_witness(leaf, child_sigs, np.random.default_rng(7000 + c), xmss=signer)
        for k in range(2):  # (the second run is the warm one)
            pr = lm.Prover(ctx)
            t0 = time.perf_counter()
            vm.prove_execution_vm(ctx, pr, leaf, pi, wit, builder)
            t1 = time.perf_counter()
        leaf_ms.append(1e3 * (t1 - t0))
        raw, claim, stmt = capi.verify_execution_raw(dict(inst, public_input=pi), pr, builder, with_statement=True)   # the library's verifier accepts the child
        children.append((raw, claim, wv.parse_raw_proof(pr.proof())[1], stmt, pi))
        assert n_vars in (None, claim.num_variables)
        n_vars = claim.num_variables
    cfg = lm.WhirConfig.new(builder, n_vars).to_dict()
    bc = wv.build_program(cfg, n_children, log_size=19 if args.scale_log == 0 else None, statement=wv.Statement(children[0][3], children[0][1], public_input_len=8), air=True, head=True, evaluators=True)
    S = bc.info["shape"]
    t0 = time.perf_counter()
    pi, wit, _ = wv.build_witness(bc, children)
    hints_ms = 1e3 * (time.perf_counter() - t0)
    phases, info = [], None
    for k in range(args.warmup + args.steps):
        if k == args.warmup:
            ctx.sync()
            t0 = time.perf_counter()
        pr = lm.Prover(ctx)
        ph = (lm.capi.C.c_double * 3)()
        ri = vm.VmRunInfo()
        pia = np.ascontiguousarray(pi, dtype=np.uint32)
        rc = ctx.lib.lmh_prove_execution_vm_info(ctx.h, pr.h, bc.handle(), pia.ctypes.data, pia.size, lm.capi.C.byref(wit.c), lm.capi.C.byref(builder), 0, ph,
                                                 lm.capi.C.byref(ri))
        if rc != 0:
            raise SystemExit("whir-recursion: " + ctx.lib.lm_last_error().decode())
        pr.proof_pruned()
        if k >= args.warmup:
            phases.append(list(ph))
        info = ri.to_dict()
    ctx.sync()
    dt = (time.perf_counter() - t0) / args.steps
    ex = vm.execute(bc, pi, wit, ctx=ctx)
    ww = dict(inst, log_bytecode=bc.log_size, ending_pc=bc.ending_pc, bytecode_hash=bc.hash(), bytecode=bc.multilinear, public_input=pi)
    ok, err = lm.verify_execution(ww, pr.proof_bytes(compressed=True), builder, compressed=True)
    stages = pr.stage_times()
    ph = np.asarray(phases).mean(axis=0)
    # the same terms as tools/recursion_shape.py counts for the reference's program (per child)
    derived = None
    try:
        d = recursion_shape()["per_child"]
        t = d["terms"]
        derived = dict(poseidon_merkle=t["poseidon.merkle"], poseidon_pow=t["poseidon.pow"], ext_leaf_folds=t["ext.leaf_folds"], ext_eq_tables=t["ext.eq_tables"],
                       ext_query_eq=t["ext.query_eq"], ext_final_poly_evals=t["ext.final_poly_evals"], ext_ood=t["ext.ood"],
                       ext_combination_dots=t["ext.combination_dots"], ext_sumcheck_verify=t["ext.sumcheck_verify"],
                       whole_verifier_poseidon_calls=d["poseidon_calls"], whole_verifier_extension_rows=d["extension_rows"], whole_verifier_cycles=d["cycles"],
                       child_stacked_n_vars=recursion_shape()["child"]["stacked_n_vars"])
    except Exception as e:  # noqa: BLE001 — the derivation is a side note
        derived = {"error": repr(e)}
    out = {
        "metric": "recursion_root_steps_per_sec (recursion() of the in-VM verifier on every child)", "value": 1.0 / dt, "unit": "root steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 (KoalaBear Montgomery, 31-bit modular)", "data": "synthetic",
        "config": {"workload": f"recursion --n {n_children} --log-inv-rate {rate} (BASELINE configs[3]), the root step as far as the recursion program is assembled: "
                               f"`recursion()` of the reference's in-VM verifier WHOLE (recursion.py:48-787 + zkdsl_implem/whir.py: GKR quotient, logup statement, batched AIR sumcheck with evaluate_air_constraints, PCS statement, whir_open) on {n_children} GENUINE child proofs of {child_sigs} real signatures each "
                               f"(stacked 2^{n_vars}, queries {S.queries}, Merkle heights {S.height}) WITH the assembly of its statement ({S.statement.n_values} claimed "
                               f"evaluations read from the children's raw transcripts, recursion.py:469-518, 534-652); main.py's recursion branch around it (public-key partition, "
                               f"bytecode-claim reduction, type-2) is NOT in the program: the child's public input and the bytecode value (a hint in the reference too) enter through the claims buffer"
                               + ("" if args.scale_log == 0 else f" [children SCALED DOWN by 2^{args.scale_log}]"),
                   "source_sha": source_sha()},
        "root": {"cycles": ex.n_cycles, "poseidon_calls": ex.n_poseidon_calls, "extension_rows": ex.n_extension_rows, "memory_words": ex.memory_len, **ex.counts,
                 "bytecode_log_size": bc.log_size, "instructions": bc.info.get("n_instructions"), "frames": bc.info["frames"], "main_frame": bc.info["main_frame_size"],
                 "per_child": {"cycles": ex.n_cycles // n_children, "poseidon_calls": ex.n_poseidon_calls // n_children, "extension_rows": ex.n_extension_rows // n_children},
                 "opening_alone_expected_from_parameters": wv.expected_counts(S), "proof_accepted": bool(ok), "verifier_message": err,
                 "proof_size_kib": pr.proof_size_fe() * 31 / 8192.0},
        "derived_by_counting_per_child": derived,
        "stages_ms": {"hints (host: claims, transcripts, opening blobs)": hints_ms, "Witness generation: Executing bytecode": float(ph[0]),
                      "Witness generation: Building execution trace": float(ph[1]), "prove_execution": float(ph[2]), **stages},
        "children": {"n": n_children, "signatures_each": child_sigs, "prove_ms_each": leaf_ms, "definition": "lmh_prove_execution_vm of one leaf at this rate (warm)"},
        "recursion_n4": {"leaves_plus_root_ms": float(sum(leaf_ms) + 1e3 * dt),
                         "reference": "README.md:60: 1.02 s for the 4 -> 1 step alone on an M4 Max (the compiled verifier incl. main.py's recursion branch; other lowering, "
                                      "other machine)"},
        **info,
    }
    if args.equal_oracle:
        from tests import synth_witness
        run = ob.VmRun(orc, bc, pi, wit)
        wo = run.trace(rate)
        ob.set_threads(orc, min(16, len(os.sched_getaffinity(0))))
        t0 = time.time()
        ref = ob.prove_execution(orc, wo, synth_witness.header(wo), ob.whir_builder(log_inv_rate=rate, soundness=ob.CAPACITY if capacity else ob.JOHNSON))
        out["config"]["proof_equals_oracle_prover"] = bool(np.array_equal(pr.proof(), ref))
        out["config"]["oracle_vm_equals_device_vm"] = bool(ex.n_cycles == run.pcs.size and np.array_equal(ex.memory(), run.memory) and np.array_equal(ex.pcs(), run.pcs))
        out["config"]["oracle_prover_s"] = time.time() - t0
    print(json.dumps(out), flush=True)
    if not ok or not out.get("vm_on_device"):
        raise SystemExit("whir-recursion: " + (err or "the (child, query) loops did not all run on the device: " + str(out.get("fallback_reason"))))


def oracle_builder(ob, w):
    """the workload's WHIR parameters in the oracle's format (checker only)"""
    return ob.whir_builder(log_inv_rate=w["log_inv_rate"], soundness=ob.CAPACITY if w["capacity"] else ob.JOHNSON)


def oracle_witness(orc, ob, w):
    """the full host-side witness of the default workload, from the ORACLE's runner and get_execution_trace (checker only)"""
    v = w["vm"]
    return ob.VmRun(orc, v["bc"], v["pi"], v["wit"]).trace(w["log_inv_rate"])


def pin_witness(w):
    """--host-resident: pinned host copies of every committed column / memory image, re-uploaded at the start of each step
    (what a node pays when the trace builder leaves the witness in host memory): (device ptr, pinned tensor) pairs."""
    import torch
    pairs = []

    def walk(x):
        if hasattr(x, "ptr") and hasattr(x, "n_words"):
            t = torch.from_numpy(x.download().view(np.int32)).pin_memory()
            pairs.append((x.ptr, t))
        elif isinstance(x, (list, tuple)):
            for y in x:
                walk(y)

    walk(w["keep"])
    return pairs


def run_step(ctx, lm, w, whole_node=None, phases=None):
    """One step.  whole_node (default: whenever the workload carries its program): lmh_aggregate_type_1 — input assembly, VM run, trace,
    proof; else lmh_prove_execution from the trace resident in HBM.  phases: list that receives [inputs_ms, vm_ms, trace_ms, prove_ms]."""
    if whole_node is None:
        whole_node = "vm" in w
    pr = lm.Prover(ctx)
    if whole_node:
        from leanmultisig_amd import vm
        v = w["vm"]
        # aggregate_type_1 whole (type_1_aggregation.rs:206-377): sort + dedup, the three input hashes, the hint map, then prove_execution
        t, info = vm.aggregate_type_1(ctx, pr, v["bc"], v["raw"], v["message"], v["slot"], w["lm_builder"], n_threads=w.get("vm_threads", 0))
        if phases is not None:
            phases.append(t)
        w["vm_run_info"] = info.to_dict()
        return pr
    for dptr, t in w.get("pinned", ()):  # PCIe-inclusive mode: host -> HBM copies are part of the step
        ctx._check(ctx.lib.lm_upload_async(ctx.h, dptr, t.data_ptr(), t.numel()))
    pr.prove_execution(w["tr"], w["cfg"])
    return pr


def step_root(pr):
    """commitment root = transcript words 6..14 (after the 6 dimension words, prove_execution.rs:50-63, commit.rs:87)"""
    blob = pr.proof()
    return blob[1 + 6:1 + 14]


def cpu_baseline(orc, ob, ctx=None, witness="xmss", log_scale=0):
    """The oracle (scalar C++ restatement of the reference algorithm; the prover's data-parallel loops — LDE, Merkle levels, sumcheck
    rounds, folds — are OpenMP loops over the host cores, as the reference's are rayon loops) on the SAME step at FULL size (log_scale = 0,
    the default since round 5: the aggregation program on 1550 real signatures — VM run (sequential restatement), get_execution_trace,
    prove_execution with 124-bit parameters; 1-2 minutes of CPU); --cpu-baseline-scale-log k takes 1550 / 2^k signatures and scales linearly.
    16 threads at most: the oracle's loops are fine-grained and stop scaling there (measured on the 256-thread host of the GPU box: 8 threads
    5.6 s, 16: 3.7 s, 32: 3.8 s, 64: 4.6 s, 256: 67 s at 1/16)."""
    from tests import synth_witness
    rng = np.random.default_rng(1)
    sh = log_scale
    want = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = ob.set_threads(orc, min(want, 16))
    if witness == "xmss":
        from leanmultisig_amd.programs import xmss_aggregate as xa
        n = N_SIGS >> sh
        bc = xa.build_program(19 if sh == 0 else None)
        signer = xa.Xmss(compress=(lambda x: ctx.poseidon16(x, compress=True)) if ctx is not None else None)
        pi, wit, _ = xa.build_witness(bc, n, rng, xmss=signer)
        t0 = time.time()
        w = ob.VmRun(orc, bc, pi, wit).trace(1)
        t_vm = time.time() - t0
        what = (f"the aggregation program on {n} real XMSS signatures ({'the full step' if sh == 0 else f'1/{1 << sh} of the step'}; "
                f"oracle VM + get_execution_trace {t_vm:.1f} s, 1 thread)")
    else:
        w = synth_witness.build(orc, rng, n_calls=(N_SIGS * 167) >> sh, n_blocks=4096 >> sh, log_exec=20 - sh, log_pos=18 - sh, log_ext=8,
                                log_memory=20 - sh, log_bytecode=19 - sh, fill_rows=None)
        t_vm = 0.0
        what = f"a consistent synthetic trace 1/{1 << sh} of the step"
    lr = w["log_rows"]
    t0 = time.time()
    ob.prove_execution(orc, w, synth_witness.header(w), None)
    dt = time.time() - t0
    est_full = (dt + t_vm) * (1 << sh)
    return dict(value=N_SIGS / est_full, unit="xmss_sigs/s", cores=cores, kind="port", seconds=dt + t_vm, scale=1 << sh,
                sample=f"oracle VM run + prove_execution (commit, logup GKR, AIR sumcheck, WHIR open; 124-bit parameters) on {what}: "
                       f"tables 2^{lr[0]}/2^{lr[2]}/2^{lr[1]}, memory 2^{w['log_memory']}, "
                       f"prove {dt:.1f} s on {cores} OpenMP threads; runs AFTER the timed region (rank 0, N = 1): it is most of this run's wall clock and none of `value`"
                       + ("" if sh == 0 else f", scaled x{1 << sh}; the fixed-size PoW searches are over-counted by the scaling"))


def source_sha():
    """sha256 over the DEVICE sources (csrc/*.hip, *.h, *.inc — not the host-only csrc/host/) the profiles under profiles/ were
    taken from: a counter file recorded for other kernels is refused (its traffic / instruction counts would silently describe
    kernels that no longer exist)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "leanmultisig_amd", "csrc")
    for f in sorted(os.listdir(base)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(base, f), "rb").read())
    return h.hexdigest()[:16]


def load_profile(name, sha):
    """profiles/<name> if it was recorded for the current kernel sources, else (None, reason)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"profiles/{name} absent"
    j = json.load(open(path))
    if j.get("source_sha") != sha:
        return None, f"profiles/{name} was recorded for sources {j.get('source_sha')} != current {sha}: refused as stale"
    return j, None


VALU_PEAK_T = 256 * 4 * 32 * 2.4e9 / 1e12  # CUs x SIMD-32 units x lanes per cycle x clock = 78.6 T lane-ops/s (MI355X_MICROARCH.md)
PROFILE_TAG = "r06"  # profiles/<tag>_{pmc,valu}_bench.json, <tag>_isa_mix.json: counter / ISA summaries of the current kernel sources (sha-guarded)


def sponge_ceiling():
    """issue-bound ceiling of the one-lane permutation in G permutations/s and where it comes from: the static VALU count and issue weight of
    ONE straight-line permutation (tools/ubench/one_perm.hip through tools/isa_mix.py --all) of the CURRENT sources; without such a file
    the figure of the sources it was last measured for is used and labelled"""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_isa_mix.json")))
        k = j["kernels"]["poseidon16_permute_one_lane"]
        stale = "" if j.get("source_sha") == source_sha() else f" [recorded for sources {j.get('source_sha')}]"
        return VALU_PEAK_T * 1e3 / (k["valu"] * k["issue_cycle_weight"]), f"{k['valu']} VALU instructions per permutation x issue weight {k['issue_cycle_weight']} (profiles/{PROFILE_TAG}_isa_mix.json{stale})"
    except Exception:  # noqa: BLE001
        return VALU_PEAK_T * 1e3 / (5626 * 1.603), "5626 VALU instructions per permutation x issue weight 1.603 (static count of the round-5 sources; no ISA summary found)"
KB_MUL_T, MAD_T = 5.65, 34.3  # measured: one Montgomery product, v_mad_u64_u32 (tools/ubench/int_rates.hip, profiles/r03_int_rates.txt)


_XBUF = {}


def exchange_step(root, proof_words, device):
    """The exchange of the sharded path, once per step: all-gather of the commitment roots (8 words) and of
    the pruned proofs.  Proof lengths differ by a few words: a length word + zero padding to a fixed capacity.  One persistent
    pinned staging buffer, one persistent send and one persistent receive tensor on `device`; `all_gather_into_tensor` writes the
    receive tensor in place and NOTHING is downloaded here — the rank that runs the 8 -> 1 recursion step reads it (`.cpu()`),
    the others never do.  One rank: nothing to exchange — the packed host buffer is returned as it is (no collective, no device
    round trip; the same code path whether launched directly or through torch.distributed.run with one process).
    Returns a (world, 9 + capacity) int32 tensor."""
    import torch
    import torch.distributed as dist
    cap = 1 << 17
    assert proof_words.size < cap
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if "host" not in _XBUF:  # zeroed once; only the words behind a shorter proof are cleared again
        host = torch.zeros(8 + 1 + cap, dtype=torch.int32)
        _XBUF["host"] = host.pin_memory() if torch.cuda.is_available() else host
        _XBUF["len"] = 0
        if world > 1:
            _XBUF["send"] = torch.zeros(8 + 1 + cap, dtype=torch.int32, device=device)
            _XBUF["recv"] = torch.zeros(world * (8 + 1 + cap), dtype=torch.int32, device=device)
    t, buf = _XBUF["host"], _XBUF["host"].numpy()
    n = int(proof_words.size)
    buf[:8] = np.asarray(root, dtype=np.uint32).view(np.int32)
    buf[8] = n
    buf[9:9 + n] = np.asarray(proof_words, dtype=np.uint32).view(np.int32)
    if _XBUF["len"] > n:
        buf[9 + n:9 + _XBUF["len"]] = 0
    _XBUF["len"] = n
    if world == 1:
        return t.view(1, -1)
    _XBUF["send"].copy_(t, non_blocking=True)
    dist.all_gather_into_tensor(_XBUF["recv"], _XBUF["send"])
    return _XBUF["recv"].view(world, -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-scale-log", type=int, default=0, help="cpu_baseline on 1550 / 2^k signatures, scaled (default 0: the full step, 1-2 min)")
    ap.add_argument("--total-signatures", type=int, default=0,
                    help="signatures of the whole job, partitioned into one contiguous leaf per rank (signer_ranges); default 1550 x ranks = BASELINE "
                         "configs[1] per GPU / configs[4] on 8")
    ap.add_argument("--no-whole-node", action="store_true", help="skip the whole_node leg (counter passes: only the timed region's proofs run)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="side measurement after the timed region (N = 1 only): independent proofs in flight on the GPU (one host "
                         "thread + HIP stream each).  0 = default: 10, fewer if the host has less than 2 hardware threads per "
                         "prover thread; 1 = skip")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N > 1 path on a single-GPU box "
                         "together with LM_BENCH_SINGLE_DEVICE=1)")
    ap.add_argument("--equal-oracle", action="store_true",
                    help="after the timed region: run the oracle PROVER (CPU, ~1-2 min at full size) on the same witness and report "
                         "whether the device proof equals its proof word for word (config.proof_equals_oracle_prover)")
    ap.add_argument("--host-resident", action="store_true",
                    help="re-upload the whole witness from pinned host memory in every step (PCIe-inclusive rate)")
    ap.add_argument("--log-inv-rate", type=int, default=1, help="WHIR rate 1/2^k (1 = BASELINE configs[1], 2 = configs[2])")
    ap.add_argument("--soundness", choices=["johnson", "capacity"], default="johnson",
                    help="capacity = the reference's `prox-gaps-conjecture` feature (lean_prover/src/lib.rs:39-43)")
    ap.add_argument("--witness", choices=["xmss", "synthetic"], default="xmss",
                    help="xmss (default): real XMSS signatures, the hash calls / copies / cycle count of the aggregation program "
                         "(tests/xmss_witness.py); synthetic: round 1's straight-line program of Poseidon calls on random inputs "
                         "(4096 operand blocks, 75 %% padding rows)")
    ap.add_argument("--scale-log", type=int, default=0, help="shrink the workload by 2^k (default 0 = config 2)")
    ap.add_argument("--shape", choices=["xmss", "recursion", "whir-recursion"], default="xmss",
                    help="xmss = BASELINE configs[1]/[2] (the metric); recursion = configs[3] stand-in: ExtensionOp table 2^19, "
                         "Poseidon 2^16, execution 2^19 — derived by tools/recursion_shape.py (side measurement, reported as proofs/s)")
    ap.add_argument("--verify", action="store_true", help="check the last proof with the oracle's verify_execution (untimed)")
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel HIP-event table of one extra step")
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this driver
    # a prover uses 4 streams (main + one per AIR table); the runtime's default of 4 hardware queues is shared by all
    # provers of the process in the `inflight` side measurement (10 provers: 105 k -> 112 k signatures/s with 16)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("LM_BENCH_SINGLE_DEVICE"):  # test rig: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")

    import leanmultisig_amd as lm
    vm_path = args.shape == "xmss" and args.witness == "xmss"
    ob = orc = None
    if not vm_path or args.verify or args.equal_oracle or (not args.no_cpu_baseline and world == 1):
        # the oracle is the CHECKER: --verify / --equal-oracle, the cpu_baseline leg and the synthetic side workloads' generator
        # (all outside the timed regions); the default workload itself never touches it
        from tests import oracle_binding as ob
        orc = ob.load()
    hw = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ctx = lm.Context(local_rank)
    capacity = args.soundness == "capacity"
    if args.shape == "whir-recursion":  # side measurement with its own JSON line (N = 1)
        assert world == 1
        return whir_recursion_bench(ctx, lm, args, ob, orc)
    # the job's signer set is cut into one contiguous leaf per rank; every rank signs its own leaf's keys (rank-seeded)
    total = args.total_signatures or max(2, N_SIGS >> args.scale_log) * world
    leaf_lo, leaf_hi = signer_ranges(total, world)[rank]
    if vm_path:
        w = build_vm_workload(ctx, np.random.default_rng(1000 + rank * 64), max(2, leaf_hi - leaf_lo), args.log_inv_rate, capacity,
                              log_bytecode=19 if args.scale_log == 0 else None)
    else:
        w = build_workload(ctx, orc, ob, np.random.default_rng(1000 + rank * 64), args.scale_log, args.log_inv_rate, args.shape, capacity, args.witness)
    if args.host_resident:
        assert not vm_path, "--host-resident applies to the synthetic witnesses; the default workload reports whole_node (hints -> proof)"
        w["pinned"] = pin_witness(w)

    # ---- the timed region: K steps, one step = ONE leaf of 1550 signatures on this rank's GPU from hints to proof, which is what
    # the reference's metric times (n_xmss / mean elapsed of one aggregate_type_1, rec_aggregation/src/benchmark.rs:397-431).
    # N ranks prove N independent leaves (weak scaling) and exchange roots + pruned proofs after every step.
    dominant = "k_air_round"
    # the HBM-bound launches: the two-rounds-per-pass product sumcheck of the WHIR opening, the GKR steps on >= 2^20 outputs
    # (recorded as k_gkr_step_big) and the single pass that writes the weight polynomial: live GB/s lines
    hbm_kernels = ("k_fold2_round", "k_prod_round2", "k_gkr_step_big", "k_weights_init")
    if vm_path:
        # sequential parts of the VM + the host pool of a batch the device hands back: this rank's share of the CPUs the cgroup grants
        w["vm_threads"] = max(2, min(128, 4 * effective_cpus()) // world)
    for _ in range(args.warmup):
        pr = run_step(ctx, lm, w)
        exchange_step(step_root(pr), pr.proof_pruned(), device)
    ctx.sync()
    ctx.profile_select(",".join((dominant, "k_leaf_sponge") + hbm_kernels))
    ctx.wait_log(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    gathered = None
    phases, stage_acc, step_s = [], {}, []
    for _ in range(args.steps):
        ts = time.perf_counter()
        pr = run_step(ctx, lm, w, phases=phases)
        gathered = exchange_step(step_root(pr), pr.proof_pruned(), device)
        step_s.append(time.perf_counter() - ts)
        for k, v in pr.stage_times().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        if os.environ.get("LM_BENCH_DEBUG"):
            import ctypes
            print(f"# step {1e3 * step_s[-1]:.2f} ms" + (" = " + " + ".join(f"{x:.2f}" for x in phases[-1]) if phases else "") +
                  f" on cpu {ctypes.CDLL(None).sched_getcpu()}", file=sys.stderr)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    waits = ctx.wait_log_read()
    ctx.wait_log(False)
    # the host side of every rank (N > 1): what an 8-GPU node needs from its CPUs.  host_busy = the part of a step the prover thread is NOT
    # waiting for the device (transcript, launch calls, the VM's sequential parts); cpu = process CPU time per step, all threads (the wait
    # is a spin: it counts); jitter = max - min of the ranks' step times, step by step
    rank_rows = None
    if world > 1:
        mine = dict(rank=rank, step_ms=[1e3 * x for x in step_s], waiting_ms_per_step=float(waits.sum()) / 1e3 / args.steps if waits.size else None,
                    cpu_ms_per_step=1e3 * cpu_s / args.steps)
        rank_rows = [None] * world
        dist.all_gather_object(rank_rows, mine)
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    n_launch, k_ms = ctx.profile_read(dominant)
    k_busy_ms = ctx.profile_busy_ms()  # the family's launches overlap (three AIR sessions on three streams): time with >= 1 of them running
    sponge_n, sponge_ms = ctx.profile_read("k_leaf_sponge")
    sponge_perms = int(ctx.lib.lm_profile_read_bytes(ctx.h, b"k_leaf_sponge"))  # (this kernel's "bytes" are permutations, lm_commit.hip)
    hbm_live = {k: ctx.profile_read(k) + (int(ctx.lib.lm_profile_read_bytes(ctx.h, k.encode())),) for k in hbm_kernels}
    ctx.profile_select(None)
    assert gathered.shape[0] == world
    if rank == 0:  # the rank that would run the recursion step is the only one that reads the gathered proofs
        g = gathered.cpu().numpy()
        assert int(g[0, 8]) == pr.proof_pruned().size and all(int(g[r, 8]) > 0 for r in range(world))

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        ww = w["w"]
        sigs = ww.get("n_sigs", ((N_SIGS * 167) >> args.scale_log) // 167)  # signatures per proof (--scale-log shrinks the leaf)
        value = (total if vm_path else sigs * world) / (dt / args.steps)  # the signatures of ALL leaves / the latency of one step
        sha = source_sha()
        # dominant kernel family: k_air_round (Poseidon16 / execution / extension_op constraint evaluation).
        # Algorithmic bytes per step: every column value of every sumcheck round is read once:
        # round 0 on base words (4 B), round r >= 1 on EF (20 B) over the 2^(log_rows - r) rows, (n_columns + n_shift) columns.
        # With the active prefix (non_padded_n_rows) a round reads the pairs that contain an active row plus one padding pair.
        alg_bytes = 0
        for t, ncols in ((0, 22), (1, 42), (2, 109)):
            lr = w["w"]["log_rows"][t]
            n_act = int(w["tr"].tables[t].non_padded_n_rows) or (1 << lr)
            for r in range(lr):
                full = 1 << (lr - r)
                pairs = (-(-n_act // (1 << r)) + 1) // 2
                rows = 2 * pairs + 2 if pairs + 1 <= full // 2 else full
                alg_bytes += ncols * (4 if r == 0 else 20) * rows
        achieved = alg_bytes * args.steps / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # counters of the same kernels (rocprofv3 --pmc passes of `bench.py --inflight 1`, summarised under profiles/ together
        # with the sha of the kernel sources they were taken from)
        traffic, alu, notes = None, None, []
        full = args.scale_log == 0 and args.shape == "xmss" and args.log_inv_rate == 1 and not capacity
        pmc, why = load_profile(PROFILE_TAG + "_pmc_bench.json", sha) if full else (None, "counter profiles are of the default workload")
        if pmc and dominant in pmc["per_step"] and n_launch:
            traffic = pmc["per_step"][dominant]["hbm_bytes"] / pmc["per_step"][dominant]["launches"]
        elif why:
            notes.append(why)
        valu, why = load_profile(PROFILE_TAG + "_valu_bench.json", sha) if full else (None, None)
        if valu and dominant in valu["per_proof"] and k_ms > 0:
            jv = valu["per_proof"][dominant]
            lane_ops = jv["valu_wave_insts"] * 64
            ach = lane_ops * args.steps / (k_ms * 1e-3) / 1e12
            wf = jv.get("issue_cycle_weight")  # issue cycles per instruction / 2, from the kernel's ISA mix (tools/isa_mix.py)
            alu = {"unit": "T VALU lane-instructions/s", "achieved": ach, "peak": VALU_PEAK_T, "frac": ach / VALU_PEAK_T,
                   "issue_cycle_weight": wf, "frac_issue_weighted": ach * wf / VALU_PEAK_T if wf else None,
                   # the same instructions over the time during which at least one launch of the family was running: the sessions' launches
                   # overlap on the chip, so the summed launch durations count shared time once per launch
                   "busy_ms_per_step": k_busy_ms / args.steps, "summed_ms_per_step": k_ms / args.steps,
                   # and with the family's launches serialised (the counter pass of the same sources: one kernel at a time)
                   "frac_issue_weighted_serialised": jv.get("alu_frac_issue_weighted"), "serialised_ms_per_step": jv.get("kernel_ms_under_pmc"),
                   "frac_issue_weighted_over_busy_time": (lane_ops * args.steps / (k_busy_ms * 1e-3) / 1e12 * wf / VALU_PEAK_T) if wf and k_busy_ms > 0 else None,
                   "source": "SQ_INSTS_VALU of k_air_round per proof (profiles/" + PROFILE_TAG + "_valu_bench.json, rocprofv3 --pmc) x 64 lanes / "
                             "HIP-event time; peak = 256 CU x 4 SIMD-32 x 2.4 GHz; issue weight = static ISA mix with "
                             "v_mul_lo/hi_u32, v_mad_u64_u32 at 4 cycles per wave64, v_lshl_add_u64 at its measured 7.4, the rest 2 "
                             "(profiles/r03_int_rates.txt)"}
        elif why:
            notes.append(why)
        out = {
            "metric": "xmss_sigs_aggregated_per_sec", "value": value, "unit": "xmss_sigs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 (KoalaBear Montgomery, 31-bit modular)",
            "data": "synthetic",
            "config": {
                "workload": f"xmss --n-signatures {sigs} --log-inv-rate {args.log_inv_rate} (BASELINE configs[{args.log_inv_rate}]): ONE "
                            + ("prove_execution(bytecode, public_input, witness) per step — leanVM run (parallel batch on the device), trace build, proof — "
                               if vm_path else "prove_execution per step, from the execution trace to the pruned proof, ")
                            + (f"on the trace of the XMSS aggregation program (leanmultisig_amd/programs/xmss_aggregate.py, looped, hand-assembled) "
                               f"verifying {sigs} REAL signatures, executed by lmh_execute_bytecode: {ww['counts']['cycles']} cycles "
                               f"({ww['counts']['add']} ADD, {ww['counts']['mul']} MUL, {ww['counts']['deref']} DEREF, {ww['counts']['jump']} JUMP), "
                               f"{ww['counts']['poseidon']} Poseidon16 calls, {ww['counts']['extension_op']} ExtensionOp rows — "
                               if "counts" in ww else "on a consistent synthetic leanVM trace (258850 Poseidon calls on random inputs) — ")
                            + f"tables 2^{ww['log_rows'][0]}x20 / 2^{ww['log_rows'][2]}x109 / 2^{ww['log_rows'][1]}x29, memory 2^{ww['log_memory']}, "
                              f"bytecode 2^{ww['log_bytecode']}, stacked 2^{w['n_vars']}, 124-bit WHIR"
                            + (" (CapacityBound: prox-gaps-conjecture)" if capacity else "")
                            + ("" if args.scale_log == 0 else f" [SCALED DOWN by 2^{args.scale_log}]"),
                "value_definition": "WHOLE NODE: signatures of one leaf x ranks / latency of one aggregate_type_1 — from the unsorted (public key, signature) "
                                    "pairs in host memory (sort + dedup, hash_pubkeys, tweak table + hash, public-input buffer + hash, hint map; since round 5 hash_pubkeys "
                                    "and the public-input hash run on a helper thread BESIDE the VM, which takes their 16 words as late inputs: LM_INPUTS_EAGER=1 restores hashes-then-VM) through "
                                    "the VM run and the trace to the pruned proof (the reference's n_xmss / mean elapsed of one aggregate_type_1, "
                                    "rec_aggregation/src/benchmark.rs:397-431); `hot_path` = the same without the VM "
                                    "run and the trace build; `inflight` = the throughput with several independent leaves queued on the same GPU",
                "stages": ["aggregate_type_1 inputs(sort, hashes, hint map)", "leanvm_run(host head + device batch)", "trace_build(device)", "fiat_shamir_preamble", "memory/bytecode access counters", "stack+whir_commit(lde+merkle+ood)", "logup_fill", "logup_gkr", "column_evaluations",
                           "batched_air_sumcheck", "statement_assembly", "whir_open(weights+sumcheck+pow+queries)", "merkle_path_pruning",
                           "exchange(roots+pruned proofs)"],
                "missing": ["the zkDSL compiler (crates/lean_compiler): the aggregation program is assembled by hand at the ISA level, raw "
                            "signatures only (no recursion branch: that code is the in-VM WHIR verifier)"],
                "per_gpu_signatures": sigs,
                "witness": "re-uploaded from pinned host memory every step (PCIe inclusive)" if args.host_resident
                           else ("hints in host memory at the start of every step (the VM log is produced on the device)" if vm_path
                                 else "resident in HBM before the timed region"),
                "whir_config": "lmh_whir_config_new (the library's own WhirConfig::new)",
                "source_sha": sha,
            },
            # the dominant kernel family is integer-ALU bound: its roofline is the VALU issue rate (256 CUs x 4 SIMD-32 x 2.4 GHz,
            # every instruction weighted by its issue cycles); `hbm` keeps the same launches priced against HBM for reference
            "roofline": {
                "kernel": dominant, "bound": "int-alu",
                # frac := the SERIALISED figure (one kernel at a time, the counter pass of the same sources): it does not move with the order in
                # which the three AIR sessions' launches happen to overlap; the live figures (summed launch durations / busy time) are beside it
                "achieved": (alu["frac_issue_weighted_serialised"] * VALU_PEAK_T) if alu and alu.get("frac_issue_weighted_serialised") else None,
                "peak": VALU_PEAK_T, "unit": "T issue-weighted VALU lane-ops/s",
                "frac": alu.get("frac_issue_weighted_serialised") if alu else None,
                # the ALGORITHMIC view beside it (utilisation counts instruction overhead as achieved): SURVEY §8(d)'s modular multiplications
                # of the AIR sumchecks / the family's live time, against the multiply-add issue rate (details under `algorithmic`)
                "frac_algorithmic": (21e9 * (int(w["tr"].tables[2].non_padded_n_rows) or (1 << ww["log_rows"][2])) / float(1 << 18)) * args.steps / (k_ms * 1e-3) / 1e12 / MAD_T if k_ms > 0 else None,
                "frac_definition": "issue-weighted VALU lane-instructions of the k_air_round family per proof (SQ_INSTS_VALU x 64 x issue weight) / the family's "
                                   "kernel time with its launches serialised (the rocprofv3 counter pass) / 78.6 T lane-ops/s",
                "frac_live_summed": alu["frac_issue_weighted"] if alu else None,
                "frac_live_over_busy_time": alu.get("frac_issue_weighted_over_busy_time") if alu else None,
                "traffic": traffic,
                "launches": n_launch, "avg_launch_ms": (k_ms / n_launch) if n_launch else None,
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes * args.steps / n_launch if n_launch else None},
                "traffic_source": "profiles/" + PROFILE_TAG + "_pmc_bench.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                  "FETCH x2 per MI355X_MICROARCH.md); refused when recorded for other kernel sources",
                "note": "live: HIP events on the prover's streams around every k_air_round launch of the timed region (one proof "
                        "alone on the chip); instruction counts from profiles/" + PROFILE_TAG + "_valu_bench.json (SQ_INSTS_VALU per proof, sha-guarded). "
                        "The constraint evaluation is integer-ALU bound — see DESIGN.md §3.  `frac_live_summed` divides by the SUM of the launch "
                        "durations; the three AIR sessions run on three streams and their large launches overlap, so shared time is "
                        "counted once per launch: `frac_live_over_busy_time` uses the time with at least one launch running, "
                        "`frac` the counter pass (one kernel at a time)",
                "alu": alu,
                # ~20 G modular multiplications per proof for the Poseidon16 AIR sumcheck at 2^18 rows (+ ~1 G for the two
                # small tables), scaled to the active rows of this workload: the family's algorithmic rate
                "algorithmic": (lambda mm: {"modmuls_per_step": mm, "achieved_T_modmul_per_s": mm * args.steps / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None,
                                            "montgomery_product_T_per_s": KB_MUL_T, "multiply_add_T_per_s": MAD_T,
                                            "frac_of_multiply_add_rate": mm * args.steps / (k_ms * 1e-3) / 1e12 / MAD_T if k_ms > 0 else None})(
                    21e9 * (int(w["tr"].tables[2].non_padded_n_rows) or (1 << ww["log_rows"][2])) / float(1 << 18)),
                "profile_notes": notes,
            },
            "roofline_hbm": (lambda n, ms, by: {
                "kernel": "+".join(hbm_kernels), "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9 if ms > 0 else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
                "frac_of_copy_rate": by / (ms * 1e-3) / 1e9 / HBM_COPY_GBS if ms > 0 else None, "copy_rate": HBM_COPY_GBS,
                "launches": n, "avg_launch_ms": ms / n if n else None, "algorithmic_bytes_per_step": by / args.steps,
                "per_kernel": {k: {"launches": v[0] / args.steps, "ms_per_step": v[1] / args.steps, "algorithmic_GB_per_step": v[2] / args.steps / 1e9,
                                   "GB_per_s": v[2] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else None} for k, v in hbm_live.items()},
                "note": "live: HIP events around every launch of the HBM-bound kernels in the timed region; algorithmic bytes = every input "
                        "value read once + every output written once, accumulated by the library at the launch sites (lm_profile_read_bytes); "
                        "peak = 8 TB/s spec, copy_rate = the 6.3 TB/s a float4 copy reaches (MI355X_MICROARCH.md)"})(
                sum(v[0] for v in hbm_live.values()), sum(v[1] for v in hbm_live.values()), sum(v[2] for v in hbm_live.values())),
        }
        # ---- the largest single kernel: the leaf sponge of the Merkle commitments (integer ALU): live permutations / s against the issue
        # ceiling of its own instruction stream, and algorithmic modular multiplications / s (~1.6 k per permutation in
        # the reference's sparse form) against the measured rate of one Montgomery product and of the 32 x 32 -> 64 multiply-add
        if sponge_ms > 0:
            perm_s = sponge_perms / (sponge_ms * 1e-3)
            SPONGE_PERM_CEILING_G, ceiling_src = sponge_ceiling()
            out["roofline_sponge"] = {
                "kernel": "k_leaf_sponge", "bound": "int-alu", "launches": sponge_n / args.steps, "ms_per_step": sponge_ms / args.steps,
                "permutations_per_step": sponge_perms / args.steps, "achieved": perm_s / 1e9, "unit": "G Poseidon1-16 permutations/s",
                "peak": SPONGE_PERM_CEILING_G, "frac": perm_s / 1e9 / SPONGE_PERM_CEILING_G,
                "algorithmic": {"modmuls_per_permutation": 1600, "achieved_T_modmul_per_s": perm_s * 1600 / 1e12, "montgomery_product_T_per_s": KB_MUL_T,
                                "multiply_add_T_per_s": MAD_T, "frac_of_multiply_add_rate": perm_s * 1600 / 1e12 / MAD_T},
                "note": "live HIP events; peak = 78.6 T VALU lane-ops/s / (" + ceiling_src + "); a modular multiplication inside a delayed-reduction dot product costs "
                        "one multiply-add, so the algorithmic rate is priced against the multiply-add issue rate (34.3 T/s measured, "
                        "profiles/r03_int_rates.txt), the Montgomery-product rate (5.65 T/s) is shown for scale"}
        # ---- the reference's own time breakdown (tracing spans) and its NodeStats fields (benchmark.rs:50-66)
        st = np.asarray(step_s)
        out["node_stats"] = {"time_secs": float(st.mean()), "time_ci_secs": float(1.96 * st.std(ddof=1) / np.sqrt(st.size)) if st.size > 1 else 0.0,
                             "samples": int(st.size), "time_min_secs": float(st.min()), "time_max_secs": float(st.max()), "n_xmss": sigs}
        if "counts" in ww:
            out["node_stats"].update(cycles=ww["counts"]["cycles"], memory=1 << ww["log_memory"], poseidons=ww["counts"]["poseidon"],
                                     dots=ww["counts"]["extension_op"])
            if "memory_words" in ww:  # the words the run USED (the committed memory column pads them to `memory`)
                out["node_stats"]["memory_words"] = ww["memory_words"]
        # GKR layers re-run with one launch per exchange because a resident kernel never got its wave slots (lm_soft_fallbacks): 0 on a device of one's own
        out["node_stats"]["tail_fallbacks"] = int(ctx.soft_fallbacks())
        stages = {k: v / args.steps for k, v in stage_acc.items()}
        if phases:
            ph = np.asarray(phases).mean(axis=0)
            stages = {"aggregate_type_1: inputs": float(ph[0]), "Witness generation: Executing bytecode": float(ph[1]),
                      "Witness generation: Building execution trace": float(ph[2]), **stages}
            out["inputs_ms"], out["witness_ms"], out["prove_ms"] = float(ph[0]), float(ph[1] + ph[2]), float(ph[3])
        out["stages_ms"] = stages
        if vm_path:
            # where the VM's parallel batch ran in the LAST timed step: a batch the device hands back to the host pool is correct but is not
            # the path this line claims to measure — the default workload must not fall back
            out.update(w.get("vm_run_info", {}))
            if not out.get("vm_on_device") and not os.environ.get("LM_VM_HOST") and not os.environ.get("LM_BENCH_ALLOW_VM_FALLBACK"):
                print(json.dumps(out), flush=True)
                raise SystemExit("bench.py: the VM's parallel batch did NOT run on the device: " + str(out.get("fallback_reason")))
        if rank_rows:
            sm = np.asarray([r["step_ms"] for r in rank_rows])          # (ranks, steps)
            out["ranks"] = {"per_rank": [dict(rank=r["rank"], ms_per_step=float(np.mean(r["step_ms"])), waiting_ms_per_step=r["waiting_ms_per_step"],
                                               host_busy_ms_per_step=(float(np.mean(r["step_ms"])) - r["waiting_ms_per_step"]) if r["waiting_ms_per_step"] is not None else None,
                                               cpu_ms_per_step=r["cpu_ms_per_step"]) for r in rank_rows],
                            "step_jitter_ms": {"mean": float((sm.max(axis=0) - sm.min(axis=0)).mean()), "max": float((sm.max(axis=0) - sm.min(axis=0)).max())},
                            "slowest_rank_ms_per_step": float(sm.mean(axis=1).max()), "fastest_rank_ms_per_step": float(sm.mean(axis=1).min()),
                            "cpus_available": effective_cpus(),
                            "definition": "host side of the sharded job: host_busy = step - time spent in lm_wait_result (the prover thread's own work); cpu = "
                                          "process CPU time per step over all threads (the wait spins); jitter = max - min over ranks of the same step"}
        if waits.size:
            out["exchanges"] = {"per_step": waits.size / args.steps, "p50_us": float(np.percentile(waits, 50)), "p99_us": float(np.percentile(waits, 99)),
                                "mean_us": float(waits.mean()), "waiting_ms_per_step": float(waits.sum()) / 1e3 / args.steps,
                                "under_20us": float((waits < 20).sum()) / args.steps,
                                "definition": "host <-> device exchanges of the prover thread (one Fiat-Shamir step each: a launched kernel or a "
                                              "message to a resident one): how long lm_wait_result waited for the published result"}
        if vm_path and not args.no_whole_node:
            # ---- rounds 1-3's headline, kept for comparison: one lmh_prove_execution per step from the trace resident in HBM
            ctx.sync()
            k = max(5, args.steps)
            for _ in range(2):
                run_step(ctx, lm, w, whole_node=False)
            th = time.perf_counter()
            for _ in range(k):
                pr_hot = run_step(ctx, lm, w, whole_node=False)
                pr_hot.proof_pruned()
            ctx.sync()
            th = (time.perf_counter() - th) / k
            out["hot_path"] = {"value": sigs / th, "unit": "xmss_sigs/s", "ms_per_step": 1e3 * th,
                               "definition": "lmh_prove_execution per step: the reference's prove_execution from the execution trace (resident "
                                             "in HBM) to the pruned proof, without the VM run and the trace build (this rank, no exchange)",
                               "proof_equals_whole_node_proof": bool(np.array_equal(pr_hot.proof(), pr.proof()))}
            out["whole_node"] = {"value": value, "unit": "xmss_sigs/s", "ms_per_step": ms_per_step, "witness_ms": out.get("witness_ms"),
                                 "inputs_ms": float(ph[0]), "vm_run_ms": float(ph[1]), "trace_ms": float(ph[2]), "prove_ms": float(ph[3]), "host_threads": w["vm_threads"],
                                 "cpus_available": effective_cpus(), "definition": "= value (kept under its round-3 name)"}
        if args.shape == "recursion":  # side measurement: not the BASELINE metric
            lr = w["w"]["log_rows"]
            out["metric"], out["unit"], out["value"] = "recursion_shaped_proofs_per_sec", "proofs/s", world / (dt / args.steps)
            out["config"]["workload"] = (f"BASELINE configs[3] stand-in, shapes DERIVED by counting the in-VM verifier's work for 4 children of 775 signatures "
                                         f"(tools/recursion_shape.py): tables 2^{lr[0]}x20 / 2^{lr[1]}x29 / 2^{lr[2]}x109, "
                                         f"memory 2^{w['w']['log_memory']}, stacked 2^{w['n_vars']}, rate 1/{1 << args.log_inv_rate}; ADD/MUL/DEREF "
                                         "instructions, all six ExtensionOp modes, Poseidon calls")
            out["config"].pop("per_gpu_signatures")
        if args.verify:
            ok, err = ob.verify_execution(orc, w["w"], pr.proof(), oracle_builder(ob, w))
            out["config"]["proof_verified_by_oracle"] = bool(ok)
            if not ok:
                print("VERIFY FAILED:", err, file=sys.stderr)
            ok, err = lm.verify_execution(w["w"], pr.proof_bytes(compressed=True), w["lm_builder"], compressed=True)
            out["config"]["proof_verified_by_library"] = bool(ok)
            if not ok:
                print("VERIFY (lmh_verify_execution) FAILED:", err, file=sys.stderr)
        if args.equal_oracle:  # the oracle PROVER on the same witness and parameters (~1-2 minutes of CPU at full size)
            from tests import synth_witness
            ob.set_threads(orc, 16)
            full = oracle_witness(orc, ob, w) if vm_path else w["w"]
            ref = ob.prove_execution(orc, full, synth_witness.header(full), oracle_builder(ob, w))
            mine = pr.proof()
            out["config"]["proof_equals_oracle_prover"] = bool(ref.size == mine.size and np.array_equal(ref, mine))
        # Proof::proof_size_fe * F::bits() / 8192 as the reference prints it (benchmark.rs:447), Merkle paths pruned
        out["config"]["proof_KiB"] = round(pr.proof_size_fe() * 31 / (8 * 1024), 1)
        out["config"]["proof_bytes_postcard"] = len(pr.proof_bytes())
        out["config"]["proof_bytes_lz4"] = len(pr.proof_bytes(compressed=True))
        out["config"]["proof_KiB_unpruned"] = round(int(pr.proof().size) * 31 / (8 * 1024), 1)
        # ---- side measurement (N = 1): independent leaves in flight on the same GPU, one lm_ctx (stream, pools, pinned buffers,
        # host thread) each — the node-level throughput when several leaves are queued
        C = args.inflight if args.inflight > 0 else max(1, min(10, hw // 2))
        if world == 1 and C > 1:
            ctxs = [ctx] + [lm.Context(local_rank) for _ in range(C - 1)]
            if vm_path:  # whole node in flight (the headline's definition), then the hot path alone
                out["inflight"] = measure_inflight(lm, orc, ob, local_rank, ctxs, w, C, max(3, args.steps // 2), args, sigs, whole_node=True)
                if "hot_path" in out:
                    out["hot_path"]["inflight"] = measure_inflight(lm, orc, ob, local_rank, ctxs, w, C, max(3, args.steps // 2), args, sigs)
            else:
                out["inflight"] = measure_inflight(lm, orc, ob, local_rank, ctxs, w, C, max(3, args.steps // 2), args, sigs)
        if not args.no_cpu_baseline and world == 1 and args.shape == "xmss":  # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(orc, ob, ctx, args.witness, log_scale=max(args.cpu_baseline_scale_log, args.scale_log))
        print(json.dumps(out), flush=True)
        if args.profile_all:
            ctx.profile_select("*")
            run_step(ctx, lm, w)
            ctx.sync()
            for k in ctx.profile_names():
                cnt, ms = ctx.profile_read(k)
                if cnt:
                    print(f"# {k:24s} launches {cnt:5d} total {ms:9.3f} ms", file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()


def effective_cpus():
    """CPUs this process may use: hardware threads, capped by the cgroup quota (the GPU boxes run under cpu.max = 16 CPUs)"""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return n


def measure_inflight(lm, orc, ob, local_rank, ctxs, w0, C, steps, args, sigs, whole_node=False):
    """C provers (own lm_ctx each) prove `steps` leaves each, concurrently; returns the aggregate rate.  whole_node: every step is
    lmh_prove_execution_vm (VM run + trace + proof): the leaves' VM runs share the host thread pool, their proofs the GPU."""
    import threading
    import torch
    # VM threads per leaf when several leaves run their VMs side by side (each on its own leased pool): half of the CPUs this
    # process may use, split between the leaves that are in their VM phase at a time (measured: 8 of a 16-CPU quota)
    vm_threads = int(os.environ.get("LM_BENCH_VM_THREADS", "0")) or max(2, effective_cpus() // 2)
    if whole_node:
        ws = [w0] * C
    elif "vm" in w0:  # the same leaf on every context: own VM run and device trace each
        from leanmultisig_amd import vm
        ws = [w0]
        for c in range(1, C):
            v = w0["vm"]
            ex = vm.execute(v["bc"], v["pi"], v["wit"])
            dt = vm.DeviceTrace(ctxs[c], v["bc"], ex, v["pi"], w0["log_inv_rate"])
            ws.append(dict(w0, tr=dt.view, keep=[dt, ex]))
    else:
        ws = [w0] + [build_workload(ctxs[c], orc, ob, np.random.default_rng(2000 + c), args.scale_log, args.log_inv_rate, args.shape,
                                    args.soundness == "capacity", args.witness) for c in range(1, C)]
    ws = [dict(x, vm_threads=vm_threads) for x in ws]
    for c in range(C):
        run_step(ctxs[c], lm, ws[c], whole_node=whole_node)
        ctxs[c].sync()
    errors = []
    start = threading.Barrier(C + 1)

    def prover_thread(c):
        try:
            torch.cuda.set_device(local_rank)  # the HIP device is per host thread
            start.wait()
            for _ in range(steps):
                run_step(ctxs[c], lm, ws[c], whole_node=whole_node).proof_pruned()
            ctxs[c].sync()
        except Exception as e:  # noqa: BLE001 — reported after the join
            errors.append(e)

    threads = [threading.Thread(target=prover_thread, args=(c,)) for c in range(C)]
    for th in threads:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    if errors:
        raise errors[0]
    out = {"proofs_in_flight": C, "value": sigs * C * steps / dt, "unit": "xmss_sigs/s", "ms_per_proof": 1e3 * dt / (C * steps),
           "proofs": C * steps}
    if whole_node:
        out["vm_threads_per_leaf"] = vm_threads
        out["cpus_available"] = effective_cpus()
    return out


if __name__ == "__main__":
    main()
