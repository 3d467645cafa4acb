"""Exploration: time the commit kernels at the config-2 shape (2^26 stacked words, 128 cols, rate 1/2)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import leanmultisig_amd as lm

n_vars = int(os.environ.get("NV", 26)); fold = 7; rate = int(os.environ.get("RATE", 1))
ctx = lm.Context(0)
rng = np.random.default_rng(0)
n = 1 << n_vars
actual = int(os.environ.get("ACTUAL", 51 << (n_vars - 6)))
ev = rng.integers(0, 0x7F000001, size=n, dtype=np.uint32); ev[actual:] = 0
d = ctx.to_device(ev)
kern = ["k_ntt_pass", "k_leaf_sponge", "k_compress_layer"]
for it in range(3):
    ctx.profile_select("*")
    t0 = time.time()
    tree = ctx.commit(d, False, n_vars, fold, rate, actual_len=actual)
    ctx.sync(); t1 = time.time()
    print("iter", it, "wall ms", (t1 - t0) * 1e3, {k: ctx.profile_read(k) for k in kern})
    tree.free()
ctx.profile_select(None)
ts = []
for it in range(5):
    t0 = time.time(); tree = ctx.commit(d, False, n_vars, fold, rate, actual_len=actual); ctx.sync(); ts.append(time.time() - t0); tree.free()
print("unprofiled wall ms", [round(t * 1e3, 2) for t in ts])
pt = rng.integers(0, 0x7F000001, size=(n_vars, 5), dtype=np.uint32)
for it in range(3):
    t0 = time.time(); ctx.mle_eval(d, False, n_vars, pt); t1 = time.time()
    print("mle_eval ms", (t1 - t0) * 1e3)
