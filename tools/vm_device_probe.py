#!/usr/bin/env python3
"""Where does a device segment's time go?  Synthetic parallel loops (tests/test_vm_device_gpu.py's builder) with bodies made of ONE
kind of work — chained Poseidon calls, ADD chains, DEREF loads from the shared prefix, witness hints, jumps — timed through
LM_VM_TIMES=1 (the '[vm] device batch' line of csrc/host/lm_vm.cpp).  usage: LM_VM_TIMES=1 python tools/vm_device_probe.py [n_segments]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import leanmultisig_amd as lm
    from leanmultisig_amd.vm import FP, K, M, Label, Witness, execute
    from tests import test_vm_device_gpu as T
    from tests import oracle_binding as ob

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1549
    ctx = lm.Context(0)
    rng = np.random.default_rng(0)

    def poseidon_body(k):
        def body(p, L):
            p.hint_witness("block", L)
            p.poseidon16(FP(L), FP(L), FP(L + 8))
            for j in range(1, k):
                p.poseidon16(FP(L + 8 * j), FP(L), FP(L + 8 * (j + 1)))
            return 8 * (k + 2)
        return body

    def half_left_body(k):  # the chain-hash form of the XMSS program: half output, hardcoded left operand from the prefix
        def body(p, L):
            p.hint_witness("block", L)
            p.poseidon16(FP(L), FP(L), FP(L + 8), half=True, left=2)
            for j in range(1, k):
                p.poseidon16(FP(L + 8 + 4 * (j - 1)), FP(L), FP(L + 8 + 4 * j), half=True, left=2)
            return 8 + 4 * (k + 2)
        return body

    def add_body(k):
        def body(p, L):
            p.hint_witness("block", L)
            for j in range(k):
                p.add(M(L + 8 + j), K(1), M(L + 7 + j) if j else M(L))
            return 8 + k + 1
        return body

    def jump_body(k):
        def body(p, L):
            p.hint_witness("block", L)
            for j in range(k):
                p.jump(K(1), K(Label(f"j{j}")), FP(0))
                for _ in range(60):
                    p.panic()
                p.label(f"j{j}")
            return 9
        return body

    def deref_body(k):
        def body(p, L):
            p.hint_witness("block", L)
            p.add(K(0), K(3), M(L + 8))
            for j in range(k):
                p.deref(L + 8, j % 4, M(L + 9 + j))
            return 9 + k + 1
        return body

    for name, body in (("empty", add_body(0)), ("add x200", add_body(200)), ("add x400", add_body(400)), ("poseidon x50", poseidon_body(50)),
                       ("poseidon x100", poseidon_body(100)), ("half+left x100", half_left_body(100)), ("jump x100", jump_body(100)),
                       ("deref x200", deref_body(200))):
        bc = T.loop_program(body).finalize()
        w = Witness(bc, 0, T.hints_for(n, rng))
        for rep in range(3):
            t0 = time.perf_counter()
            ex = execute(bc, T.PI, w, n_threads=2, ctx=ctx, lazy=True)
            dt = time.perf_counter() - t0
            assert ex.on_device
            ex.close()
        print(f"## {name}: whole run {1e3 * dt:.2f} ms", file=sys.stderr, flush=True)
