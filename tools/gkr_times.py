import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import leanmultisig_amd as lm
ctx = lm.Context(0); rng = np.random.default_rng(0)
P = 0x7F000001
for n in (12, 16, 20, 23, 25):
    L = 1 << n
    nums = rng.integers(0, P, size=L, dtype=np.uint32)
    dens = rng.integers(0, P, size=(5, L), dtype=np.uint32)
    dn, dd = ctx.to_device(nums), ctx.to_device(dens)
    for it in range(3):
        pr = lm.Prover(ctx); ctx.sync(); t = time.perf_counter()
        pr.prove_gkr_quotient(dn, dd, n); ctx.sync(); dt = time.perf_counter() - t
    ctx.profile_select("*")
    pr = lm.Prover(ctx); pr.prove_gkr_quotient(dn, dd, n); ctx.sync()
    ks = {k: ctx.profile_read(k) for k in ["k_gkr_layer_up", "k_prefix_eq_tables", "k_gkr_round_storage", "k_gkr_fold_round", "k_gkr_reduce"]}
    ctx.profile_select(None)
    print(f"n={n}: wall {1e3*dt:.2f} ms; kernels " + " ".join(f"{k[6:]}={v[0]}x/{v[1]:.2f}ms" for k, v in ks.items()))
