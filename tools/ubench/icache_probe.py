"""Is a latency-bound launch bound by instruction fetch?  k_pow_grind is 55 KB of straight-line code executed once per wave.
Times lm_pow_grind (one launch + publish + host spin) back to back (code warm in the caches) and after a pass that streams
256 MB through the chip (code evicted, as between two PoWs of a proof).  usage: python tools/ubench/icache_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np

import leanmultisig_amd as lm

ctx = lm.Context(0)
rng = np.random.default_rng(1)
P = 0x7F000001
cap = rng.integers(0, P, size=8, dtype=np.uint32)
big = ctx.to_device(rng.integers(0, P, size=1 << 26, dtype=np.uint32))


def t_pow(bits, n, evict):
    ts = []
    for i in range(n):
        c = cap.copy()
        c[0] = i + 1
        if evict:
            out = ctx.stack_columns(1 << 26, [(big, 0, 0, 1 << 26)])  # streams 512 MB through the caches
            ctx.sync()
            out.free() if hasattr(out, "free") else None
        t0 = time.perf_counter()
        ctx.pow_grind(c, bits)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for bits in (4, 14):
    w = t_pow(bits, 60, False)
    e = t_pow(bits, 60, True)
    print(f"bits {bits:2d}: back to back median {w[0]:6.1f} us (min {w[1]:6.1f});  after streaming 256 MB: median {e[0]:6.1f} us (min {e[1]:6.1f})")
