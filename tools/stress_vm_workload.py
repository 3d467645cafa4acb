"""Repeated whole workloads in ONE process: real signatures -> hints -> device VM -> device trace -> proof, with a new program instance,
witness and WHIR configuration every time and the previous workload released in between (the pattern of the full-size GPU tests).
Every proof is verified by lmh_verify_execution.  usage: python tools/stress_vm_workload.py [iterations]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import leanmultisig_amd as lm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = lm.Context(0)
t0 = time.time()
for i in range(n):
    rate, cap = (1, False) if i % 3 == 0 else ((2, False) if i % 3 == 1 else (1, True))
    w = bench.build_vm_workload(ctx, np.random.default_rng(100 + i), bench.N_SIGS if i % 4 else 700, rate, cap)
    pr = bench.run_step(ctx, lm, w)
    ok, err = lm.verify_execution(w["w"], pr.proof_bytes(compressed=True), w["lm_builder"], compressed=True)
    assert ok, err
    print(f"iteration {i}: rate 1/{1 << rate} capacity {cap}: proof verified ({time.time() - t0:.0f} s)", flush=True)
    del w, pr
    gc.collect()
print(f"{n} workloads, every proof verified")
