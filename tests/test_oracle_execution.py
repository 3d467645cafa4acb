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


def test_mixed_program_pins_the_air_restatements(orc):
    """A witness written from the reference's EXECUTOR semantics (ADD / MUL / DEREF rows, execution/air.rs:96-112; all six
    ExtensionOp modes as exec_multi_row lays them out, extension_op/exec.rs:95-189) must be accepted by the oracle's
    restatement of the constraint systems, and single-cell corruptions of it must be rejected: the constraints are pinned by
    the semantics they encode, not only by padding rows."""
    import copy
    rng = np.random.default_rng(5)
    w = synth_witness.build_mixed(orc, rng)
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    proof = ob.prove_execution(orc, w, synth_witness.header(w), b)
    ok, err = ob.verify_execution(orc, w, proof, b)
    assert ok, err

    def outcome(mutate):
        w2 = copy.deepcopy(w)
        mutate(w2["tables"])
        try:
            pr = ob.prove_execution(orc, w2, synth_witness.header(w2), b)
        except RuntimeError as e:        # the lookups no longer balance
            assert "logup" in str(e)
            return False
        return ob.verify_execution(orc, w2, pr, b)[0]

    def flip(t, col, row):
        def m(tables):
            tables[t][col, row] ^= 1
        return m

    first_mul_row = 3          # ext table: 3 add_ee rows, then the 4-row dot_product_ee calls
    assert not outcome(flip(1, 9, first_mul_row + 2))      # computation coordinate inside a dot product
    assert not outcome(flip(1, 2, first_mul_row + 1))      # len must count down
    assert not outcome(flip(0, 22, 41))                    # nu_b of a MUL instruction (pc 41: kind = 1)
    assert not outcome(flip(0, 6, 42))                     # value_b of a DEREF (breaks the memory lookup)
