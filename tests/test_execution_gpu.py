"""GPU end-to-end: lmh_prove_execution (commit -> logup/GKR -> AIR -> WHIR open on the device) on a consistent synthetic
witness.  The proof must equal the oracle's proof word for word and be accepted by the oracle's verify_execution."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests import synth_witness

pytestmark = pytest.mark.gpu


def _device_proof(ctx, orc, w, builder):
    tr, keep = lm.make_execution_trace(ctx, w)
    n = ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))
    cfg = lm.WhirConfig.from_dict(ob.whir_config(orc, builder, n))
    pr = lm.Prover(ctx)
    pr.prove_execution(tr, cfg)
    return pr.proof()


def test_prove_execution_matches_oracle_and_verifies(ctx, orc):
    rng = np.random.default_rng(0)
    w = synth_witness.build(orc, rng, n_calls=40)
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    ref = ob.prove_execution(orc, w, synth_witness.header(w), b)
    proof = _device_proof(ctx, orc, w, b)
    ok, err = ob.verify_execution(orc, w, proof, b)
    assert ok, err
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_prove_execution_production_parameters_verifies(ctx, orc):
    """default_whir_config (124-bit, 16 PoW bits, lean_prover/src/lib.rs:22-50), different table heights, bigger program:
    too slow for the oracle PROVER, checked by the oracle VERIFIER."""
    rng = np.random.default_rng(1)
    w = synth_witness.build(orc, rng, n_calls=900, n_blocks=32, log_exec=11, log_pos=10, log_ext=8, log_memory=16, log_bytecode=10)
    proof = _device_proof(ctx, orc, w, ob.whir_builder(log_inv_rate=1))
    ok, err = ob.verify_execution(orc, w, proof, None)
    assert ok, err


def test_inconsistent_witness_is_rejected(ctx, orc):
    rng = np.random.default_rng(2)
    w = synth_witness.build(orc, rng, n_calls=40)
    w["memory_acc"] = w["memory_acc"].copy()
    w["memory_acc"][200] = int(orc.to_monty(77))  # wrong access count -> logup sum != 0
    with pytest.raises(lm.LmError):
        _device_proof(ctx, orc, w, ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60))
