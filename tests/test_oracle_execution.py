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


def test_execution_table_fill_follows_the_reference_loop(orc):
    """The restatement of get_execution_trace's main loop (trace_gen.rs:27-100) reproduces the hand-built execution table of
    the mixed program column by column — except where the reference makes a different (equally valid) choice: an operand
    that is an immediate gets address 0 and memory[0] as its value there, the zero vector and 0 in synth_witness — and the
    witness with the filled table is still proven and verified."""
    rng = np.random.default_rng(5)
    w = synth_witness.build_mixed(orc, rng)
    pcs, fps = synth_witness.vm_log(w)
    ex, got = w["tables"][0], ob.execution_table_fill(orc, pcs, fps, w["bytecode"], w["memory"])
    for c in list(range(0, 2)) + list(range(8, 24)):
        assert np.array_equal(got[c], ex[c]), c
    two = int(orc.to_monty(2))
    for k, flag_col in enumerate((11, 12, 13)):          # addr/value of operand k agree wherever it is a memory operand
        is_mem = (ex[flag_col] == 0) & (ex[14 if k == 2 else 15] == 0)   # not an immediate and not fp-relative
        if k == 1:
            is_mem |= ex[18] == two                      # DEREF reads memory at value_a + operand_b although flag_b = 1
        assert np.array_equal(got[2 + k][is_mem], ex[2 + k][is_mem]) and np.array_equal(got[5 + k][is_mem], ex[5 + k][is_mem])
        assert not got[2 + k][~is_mem].any() and (got[5 + k][~is_mem] == w["memory"][0]).all()
    w2 = synth_witness.with_execution_table(orc, w, got)
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    proof = ob.prove_execution(orc, w2, synth_witness.header(w2), b)
    ok, err = ob.verify_execution(orc, w2, proof, b)
    assert ok, err
