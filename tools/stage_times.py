"""Exploration: wall time per stage of one bench step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import leanmultisig_amd as lm
import bench
from tests import oracle_binding as ob
orc = ob.load(); ctx = lm.Context(0); rng = np.random.default_rng(0)
w = bench.build_workload(ctx, orc, ob, rng, 26, 1, 25)
for it in range(3):
    ts = []
    ctx.sync(); t = time.perf_counter()
    pr = lm.Prover(ctx)
    wit = pr.whir_commit(w["cfg"], w["d_poly"], w["actual"]); ctx.sync(); ts.append(time.perf_counter())
    pr.prove_gkr_quotient(w["d_nums"], w["d_dens"], w["gkr_log_n"]); ctx.sync(); ts.append(time.perf_counter())
    c = w["air_ch"]
    pr.prove_batched_air_sumcheck(w["air_tables"], c["alpha"], c["eq16"], c["beta"], c["eta"]); ctx.sync(); ts.append(time.perf_counter())
    pr.whir_prove(w["cfg"], w["sts"], wit, w["d_poly"]); ctx.sync(); ts.append(time.perf_counter())
    names = ["commit", "gkr", "air", "whir_open"]
    prev = t
    print("iter", it, " ".join(f"{n}={1e3*(x-p):.2f}ms" for n, x, p in zip(names, ts, [t] + ts[:-1])), f"total={1e3*(ts[-1]-t):.2f}ms")
