#!/usr/bin/env python3
"""Whole-node timing of lmh_prove_execution_vm (VM run + device trace + proof) on the hand-assembled aggregation program."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leanmultisig_amd as lm
from leanmultisig_amd import vm
from leanmultisig_amd.programs import xmss_aggregate as xa

n_sigs = int(sys.argv[1]) if len(sys.argv) > 1 else 1550
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
ctx = lm.Context(0)
t = time.time()
bc = xa.build_program(19)
print("assemble", round(time.time() - t, 2), "s", flush=True)
t = time.time()
x = xa.Xmss(compress=lambda s: ctx.poseidon16(s, compress=True))
pi, w, info = xa.build_witness(bc, n_sigs, np.random.default_rng(1), xmss=x)
print("sign + hints", round(time.time() - t, 2), "s; host poseidon:", lm.host_poseidon_backend(), flush=True)
for nt in threads:
    best = None
    for _ in range(5):
        t = time.perf_counter()
        ex = vm.execute(bc, pi, w, n_threads=nt)
        dt = (time.perf_counter() - t) * 1e3
        best = dt if best is None else min(best, dt)
    print(f"runner threads={nt}: {best:.2f} ms, cycles {ex.n_cycles}, memory {ex.memory_len}", flush=True)
b = lm.WhirBuilder.default(1)
for nt in threads:
    for rep in range(4):
        pr = lm.Prover(ctx)
        t = time.perf_counter()
        times = vm.prove_execution_vm(ctx, pr, bc, pi, w, b, n_threads=nt)
        tot = (time.perf_counter() - t) * 1e3
        print(f"whole node threads={nt}: total {tot:.2f} ms = vm {times[0]:.2f} + trace {times[1]:.2f} + prove {times[2]:.2f}", flush=True)
ww = dict(log_bytecode=bc.log_size, ending_pc=bc.ending_pc, public_input=pi, bytecode_hash=bc.hash(), bytecode=bc.multilinear)
ok, err = lm.verify_execution(ww, pr.proof_bytes(compressed=True), b, compressed=True)
print("verified:", ok, err, "proof KiB", round(pr.proof_size_fe() * 31 / 8192, 1))
