"""CPU test: the oracle's restatement of prove_execution -> verify_execution on a hand-built consistent witness
(the zkVM end-to-end test of the reference, lean_prover/src/test_zkvm.rs, without the VM interpreter)."""
import numpy as np

from tests import oracle_binding as ob
from tests import synth_witness


def test_prove_then_verify_execution(orc):
    rng = np.random.default_rng(0)
    w = synth_witness.build(orc, rng, n_calls=40)
    hdr = synth_witness.header(w)
    # reduced PoW / security so the scalar oracle finishes in seconds; same protocol flow
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    proof = ob.prove_execution(orc, w, hdr, b)
    ok, err = ob.verify_execution(orc, w, proof, b)
    assert ok, err
    bad = proof.copy()
    bad[50] ^= 1
    assert not ob.verify_execution(orc, w, bad, b)[0]
    pi2 = w["public_input"].copy()
    pi2[3] ^= 1
    assert not ob.verify_execution(orc, w, proof, b, public_input=pi2)[0]
