#!/usr/bin/env python3
"""bench.py — XMSS signatures aggregated per second on the proving hot path, 1550 signatures, WHIR rate 1/2.

One "step" = one pass of the proving hot path over one batch of synthetic input of the config-2 shape
(BASELINE.json configs[1]: `xmss --n-signatures 1550 --log-inv-rate 1`):
    stacked polynomial 2^26 words (51*2^20 non-zero)  -> WHIR commit (LDE 2^20 x 128 + Poseidon1-16 Merkle tree)
    logup vector of 2^25 (num, den) pairs              -> GKR sum-of-fractions proof
    252 sparse claims on the stacked polynomial        -> WHIR open (weights, 26 sumcheck rounds, 3 folded commitments,
                                                          PoW grinding, 370 Merkle openings), 124-bit parameters
Inputs are resident in HBM before the timed region.  Stages not yet on the device are listed in config["missing"]:
the metric is only the reference's whole-node number once that list is empty.

Multi-GPU (north_star / SURVEY.md §8(e)): independent 1550-signature leaves, one per GPU, no data-path collective; the
only exchange is an RCCL all-gather of the 8-word commitment roots at the end of each step.  scaling = "weak".

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


def exchange_roots(root_words, device):
    """All-gather of the per-rank 8-word commitment roots (the only collective of the sharded path).
    Works on any backend (nccl = RCCL on the GPUs, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(np.asarray(root_words, dtype=np.int64), device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.cpu().numpy().reshape(1, 8)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def signer_ranges(n_total, world):
    """Partition of the sorted signer set into `world` contiguous leaves."""
    base, rem = divmod(n_total, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


def build_workload(ctx, orc, ob, rng, n_vars, log_inv_rate, gkr_log_n):
    """Synthetic inputs of the config-2 shape (SURVEY.md §8 size table), uploaded once."""
    import leanmultisig_amd as lm
    n = 1 << n_vars
    actual = 51 << (n_vars - 6)
    poly = rng.integers(0, P, size=n, dtype=np.uint32)
    poly[actual:] = 0
    d_poly = ctx.to_device(poly)
    builder = ob.whir_builder(log_inv_rate=log_inv_rate)
    cfgd = ob.whir_config(orc, builder, n_vars)
    cfg = lm.WhirConfig.from_dict(cfgd)
    # 252 claims (stacked_pcs.rs:206-221: 6 + 28 + 60 + 145 + 13), as (log block size, count, #values per point)
    groups = [(n_vars - 5, 6, 1), (n_vars - 6, 14, 2), (n_vars - 14, 60, 1), (n_vars - 8, 145, 1), (n_vars - 7, 13, 1)]
    sts = []
    for k, count, per in groups:
        done = 0
        while done < count:
            pt = ob.rand_field(rng, (k, 5))
            vals = []
            used = set()
            for _ in range(min(per, count - done)):
                sel = int(rng.integers(0, max(1, actual >> k)))
                while sel in used:
                    sel = (sel + 1) % max(1, actual >> k)
                used.add(sel)
                vals.append((sel, ctx.mle_eval(d_poly.ptr + 4 * (sel << k), False, k, pt)[0]))
            sts.append(dict(point=pt, is_next=False, values=vals))
            done += len(vals)
    # logup vector: L = 16.5M active pairs padded to 2^25 (logup.rs:495-518), natural order, (0,1) padding
    L = 1 << gkr_log_n
    active = int(L * 16.5 / 32)
    nums = rng.integers(0, P, size=L, dtype=np.uint32)
    nums[active:] = 0
    dens = rng.integers(0, P, size=(5, L), dtype=np.uint32)  # SoA planes
    dens[:, active:] = 0
    dens[0, active:] = 0x01FFFFFE
    d_nums = ctx.to_device(nums)
    d_dens = ctx.to_device(dens)
    # AIR tables (SURVEY.md §8 size table): execution 2^(n-6) x 20, extension_op 2^10 x 29, poseidon16 2^(n-8) x 109.
    # The Poseidon table is a satisfiable trace (oracle trace generator, untimed setup); the other two carry random
    # columns — same arithmetic, the sumcheck does not depend on satisfiability.
    air_tables = []
    for table, lr in ((0, n_vars - 6), (2, n_vars - 8), (1, 10)):
        if table == 2:
            cols = ob.poseidon_table(orc, rng, lr)
        else:
            cols = rng.integers(0, P, size=(ob.AIR_N_COLUMNS[table], 1 << lr), dtype=np.uint32)
        bufs = [ctx.to_device(c) for c in cols]
        air_tables.append(dict(table=table, log_rows=lr, cols=bufs, eq_point=ob.rand_field(rng, (lr, 5)),
                               sum=ob.rand_field(rng, 5)))
    air_ch = dict(alpha=ob.rand_field(rng, 5), eq16=ob.rand_field(rng, (16, 5)), beta=ob.rand_field(rng, 5),
                  eta=ob.rand_field(rng, 5))
    return dict(cfg=cfg, cfgd=cfgd, builder=builder, d_poly=d_poly, actual=actual, sts=sts, d_nums=d_nums, d_dens=d_dens,
                gkr_log_n=gkr_log_n, n_vars=n_vars, poly_host=poly, air_tables=air_tables, air_ch=air_ch)


def run_step(ctx, lm, w):
    pr = lm.Prover(ctx)
    wit = pr.whir_commit(w["cfg"], w["d_poly"], w["actual"])
    root = np.empty(8, dtype=np.uint32)
    ctx.lib.lmh_witness_root(wit, root.ctypes.data)
    pr.prove_gkr_quotient(w["d_nums"], w["d_dens"], w["gkr_log_n"])
    c = w["air_ch"]
    pr.prove_batched_air_sumcheck(w["air_tables"], c["alpha"], c["eq16"], c["beta"], c["eta"])
    pr.whir_prove(w["cfg"], w["sts"], wit, w["d_poly"])
    return pr, root


def cpu_baseline(orc, ob, log_scale=6):
    """Oracle (scalar C++ port, OpenMP only inside PoW grinding) on a 1/2^log_scale sample of the same step."""
    rng = np.random.default_rng(1)
    n = 26 - log_scale
    poly = ob.rand_field(rng, 1 << n)
    actual = 51 << (n - 6)
    poly[actual:] = 0
    b = ob.whir_builder(log_inv_rate=1)
    sts = ob.random_statements(orc, rng, poly, n, n_points=12)
    nums, dens = ob.gkr_instance(orc, rng, 25 - log_scale, 16.5 / 32)
    t0 = time.time()
    ob.whir_prove(orc, b, n, poly, sts, actual_len=actual)
    ob.gkr_prove(orc, nums, dens)
    dt = time.time() - t0
    est_full = dt * (1 << log_scale)
    return dict(value=N_SIGS / est_full, unit="xmss_sigs/s", cores=os.cpu_count(), kind="port",
                sample=f"oracle WHIR commit+open (n={n}, 124-bit params, {len(sts)} statements) + GKR 2^{25 - log_scale} = "
                       f"1/{1 << log_scale} of the step, {dt:.1f}s measured, scaled x{1 << log_scale}; scalar except "
                       f"OpenMP PoW search")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-vars", type=int, default=26)
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel HIP-event table of one extra step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import leanmultisig_amd as lm
    from tests import oracle_binding as ob
    orc = ob.load()  # only for WhirConfig integers (f64 derivation stays on the caller side) and the cpu_baseline leg
    ctx = lm.Context(local_rank)
    rng = np.random.default_rng(1000 + rank)
    gkr_log_n = args.n_vars - 1
    w = build_workload(ctx, orc, ob, rng, args.n_vars, 1, gkr_log_n)

    for _ in range(args.warmup):
        run_step(ctx, lm, w)
    dominant = "k_leaf_sponge"
    ctx.profile_select(dominant)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    roots = None
    for _ in range(args.steps):
        pr, root = run_step(ctx, lm, w)
        roots = exchange_roots(root, device)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    n_launch, k_ms = ctx.profile_read(dominant)
    ctx.profile_select(None)
    assert roots is not None and roots.shape[0] == world

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = N_SIGS * world / (dt / args.steps)
        # dominant kernel: leaf sponge.  Algorithmic bytes per step: read every stored LDE word
        # once, write one 32-byte digest per row, for the 4 trees of a proof.
        cfgd = w["cfgd"]
        trees = [(args.n_vars + 1 - 7, 102)]  # base tree: 2^20 rows x 102 stored columns
        lir = 1
        nv = args.n_vars - 7
        for r in range(cfgd["n_rounds"]):
            lir = lir + (7 if r == 0 else 5) - (5 if r == 0 else 1)
            trees.append((nv + lir - 5, 160))
            nv -= 5
        alg_bytes = sum((1 << lh) * (4 * cols + 32) for lh, cols in trees)
        achieved = alg_bytes * args.steps / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "xmss_sigs_aggregated_per_sec", "value": value, "unit": "xmss_sigs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 (KoalaBear Montgomery, 31-bit modular)",
            "data": "synthetic",
            "config": {
                "workload": "xmss --n-signatures 1550 --log-inv-rate 1 (BASELINE configs[1]) — proving hot path on "
                            "synthetic trace shapes: stacked 2^26, LDE 2^20x128, logup 2^25, 252 claims, 124-bit WHIR",
                "stages": ["whir_commit(lde+merkle+ood)", "logup_gkr", "batched_air_sumcheck(execution 2^20, poseidon16 2^18, extension_op 2^10)",
                           "whir_open(weights+sumcheck+pow+queries)"],
                "missing": ["logup numerator/denominator build + 91 column evaluations", "witness generation (CPU VM)"],
                "per_gpu_signatures": N_SIGS,
            },
            "roofline": {
                "kernel": dominant, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "launches": n_launch, "avg_launch_ms": (k_ms / n_launch) if n_launch else None,
                "note": "Poseidon1-16 sponge is int-ALU bound (~1.6k modmul per 32 B hashed); HBM fraction is small by "
                        "construction — see DESIGN.md",
            },
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(orc, ob)
        print(json.dumps(out), flush=True)
        if args.profile_all:
            ctx.profile_select("*")
            run_step(ctx, lm, w)
            ctx.sync()
            names = ["k_ntt_pass", "k_leaf_sponge", "k_compress_layer", "k_weight_tables", "k_weights_accumulate",
                     "k_prod_round_base", "k_prod_round_ext", "k_sum10", "k_fold_base", "k_fold_ext", "k_pow_grind",
                     "k_mle_partial_base", "k_mle_partial_ext", "k_eq_table_small", "k_sum_partials", "k_tree_open",
                     "k_gkr_layer_up", "k_prefix_eq_tables", "k_gkr_round_storage", "k_gkr_fold_round", "k_gkr_reduce",
                     "k_air_round", "k_air_reduce", "k_air_fold_base", "k_air_fold_ext"]
            for k in names:
                cnt, ms = ctx.profile_read(k)
                if cnt:
                    print(f"# {k:24s} launches {cnt:5d} total {ms:9.3f} ms", file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
