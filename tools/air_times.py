"""Per-round wall time of the device AIR sumcheck session (lm_air_round + lm_air_bind) on random columns.
usage: python tools/air_times.py [table=2] [log_rows=18]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import leanmultisig_amd as lm
from leanmultisig_amd import capi

table = int(sys.argv[1]) if len(sys.argv) > 1 else 2
log_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 18
N_COLS = {0: 20, 1: 29, 2: 109}[table]
P = 0x7F000001
ctx = lm.Context(0)
rng = np.random.default_rng(1)
lib = ctx.lib
cols = [ctx.to_device(rng.integers(0, P, size=1 << log_rows, dtype=np.uint32)) for _ in range(N_COLS)]
ptrs = (C.c_void_p * N_COLS)(*[c.ptr for c in cols])
f = lambda *shape: np.ascontiguousarray(rng.integers(0, P, size=shape, dtype=np.uint32))
eqp, alpha, eq16, beta = f(log_rows, 5), f(5), f(16, 5), f(5)
vp = lambda a: a.ctypes.data_as(C.c_void_p)

for rep in range(3):
    h = C.c_void_p()
    rc = lib.lm_air_new(ctx.h, table, ptrs, log_rows, vp(eqp), vp(alpha), vp(eq16), vp(beta), C.byref(h))
    assert rc == 0, ctx.lib.lm_last_error()
    deg = lib.lm_air_degree(h)
    out = np.zeros(deg * 5, dtype=np.uint32)
    times = []
    ctx.sync()
    t_all = time.perf_counter()
    for r in range(log_rows):
        t = time.perf_counter()
        assert lib.lm_air_round(ctx.h, h, vp(out)) == 0, ctx.lib.lm_last_error()
        t1 = time.perf_counter()
        ch = f(5)
        assert lib.lm_air_bind(ctx.h, h, vp(ch)) == 0
        ctx.sync()
        times.append((t1 - t, time.perf_counter() - t1))
    total = time.perf_counter() - t_all
    lib.lm_air_free(ctx.h, h)
print(f"table {table} log_rows {log_rows}: total {1e3 * total:.2f} ms (incl. host rng)")
print("round ms:", " ".join(f"{1e3 * a:.3f}" for a, _ in times))
print("bind  ms:", " ".join(f"{1e3 * b:.3f}" for _, b in times))
print("checksum", int(out.sum()))
