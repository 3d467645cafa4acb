"""Exploration: per-kernel breakdown of commit + WHIR open at the config-2 shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import leanmultisig_amd as lm
from tests import oracle_binding as ob

n = int(os.environ.get("NV", 26)); rate = int(os.environ.get("RATE", 1))
orc = ob.load()
ctx = lm.Context(0)
rng = np.random.default_rng(0)
actual = 51 << (n - 6)
poly = rng.integers(0, ob.P, size=1 << n, dtype=np.uint32); poly[actual:] = 0
d_poly = ctx.to_device(poly)
b = ob.whir_builder(log_inv_rate=rate)
cfgd = ob.whir_config(orc, b, n)
cfg = lm.WhirConfig.from_dict(cfgd)
# synthetic statements: 250 eq claims on column-like blocks of 2^(n-6) .. 2^(n-8)
sts = []
for i in range(100):
    k = n - 6 - (i % 3)
    pt = ob.rand_field(rng, (k, 5))
    sel = int(rng.integers(0, (actual >> k)))
    vals = [(sel, ctx.mle_eval(d_poly.ptr + 4 * (sel << k), False, k, pt)[0])]
    if i % 2 == 0:
        sel2 = (sel + 1) % (actual >> k)
        vals.append((sel2, ctx.mle_eval(d_poly.ptr + 4 * (sel2 << k), False, k, pt)[0]))
    sts.append(dict(point=pt, is_next=False, values=vals))
kernels = ["k_ntt_pass", "k_leaf_sponge", "k_compress_layer", "k_weight_tables", "k_weights_accumulate", "k_prod_round_base",
           "k_prod_round_ext", "k_sum10", "k_fold_base", "k_fold_ext", "k_pow_grind", "k_mle_partial_base", "k_mle_partial_ext",
           "k_eq_table_small", "k_sum_partials", "k_tree_open"]
for it in range(3):
    prof = it == 2
    ctx.profile_select("*" if prof else None)
    pr = lm.Prover(ctx)
    ctx.sync(); t0 = time.time()
    wit = pr.whir_commit(cfg, d_poly, actual)
    ctx.sync(); t1 = time.time()
    pt = pr.whir_prove(cfg, sts, wit, d_poly)
    ctx.sync(); t2 = time.time()
    print(f"iter {it}: commit {1e3*(t1-t0):.2f} ms, open {1e3*(t2-t1):.2f} ms, proof words {pr.proof().size}")
    if prof:
        tot = 0
        for k in kernels:
            cnt, ms = ctx.profile_read(k)
            tot += ms
            print(f"   {k:24s} launches {cnt:5d}  total {ms:8.3f} ms")
        print("   sum of kernels", tot)
if os.environ.get("VERIFY"):
    ok, vpt, err = ob.whir_verify(orc, b, n, pr.proof(), sts)
    print("oracle verify:", ok, err)
