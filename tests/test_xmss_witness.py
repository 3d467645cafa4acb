"""SURVEY.md §8(f) rank 4, first step: real XMSS signatures (tests/xmss_py.py, restating crates/xmss) and the execution
witness of verifying them (tests/xmss_witness.py, following zkdsl_implem/xmss_aggregate.py).  CPU: the scheme round-trips
and rejects tampering; the witness has the reference's per-signature counts; the oracle PROVES it (the logup sum is zero
only if every lookup, bus entry and access counter is consistent) and both verifiers accept the proof.  GPU: the device
proof of such a witness equals the oracle's word for word."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests import synth_witness, xmss_witness
from tests.xmss_py import NUM_CHAIN_HASHES, TARGET_SUM, V, Xmss

SLOT = 0x12345678


def test_xmss_sign_verify_and_tampering(orc):
    x = Xmss(orc)
    rng = np.random.default_rng(1)
    msg = ob.rand_field(rng, 8)
    sig = x.keygen_and_sign(rng, 6, msg, SLOT, ob.rand_field)
    assert np.all(sig["encoding"].sum(axis=1) == TARGET_SUM) and np.all((0 <= sig["encoding"]) & (sig["encoding"] < 8))
    assert np.all(7 * V - sig["encoding"].sum(axis=1) == NUM_CHAIN_HASHES)   # 110 chain hashes per verification
    assert x.verify(sig, msg, SLOT).all()
    for key, idx in (("chain_tips", (2, 5, 1)), ("merkle_proof", (2, 31, 0)), ("randomness", (2, 0)), ("root", (2, 3)), ("pp", (2, 1))):
        bad = dict(sig)
        bad[key] = sig[key].copy()
        bad[key][idx] ^= 1
        ok = x.verify(bad, msg, SLOT)
        assert not ok[2] and ok[[0, 1, 3, 4, 5]].all(), key
    assert not x.verify(sig, msg, SLOT + 1).any()
    msg2 = msg.copy()
    msg2[0] ^= 1
    assert not x.verify(sig, msg2, SLOT).any()


def test_witness_counts_and_consistency(orc):
    w = xmss_witness.build(orc, np.random.default_rng(7), n_sigs=3, n_arith=45, slot=SLOT)
    c = w["counts"]
    assert c["poseidon"] == 3 * 166                                        # 2 + 110 + 22 + 32 per signature
    n_zero_chains = int((w["xmss"]["encoding"] == 7).sum())
    assert c["extension_op"] == 3 * (3 + 8) + n_zero_chains                # copy_6 / copy_5 / zeros, one copy per Merkle chunk, untouched chains
    assert c["cycles"] == c["poseidon"] + c["extension_op"] + 3 * 45
    pos = w["tables"][2]
    active = orc.from_monty_fast(pos[0]) == 1
    half, hard = orc.from_monty_fast(pos[3])[active], orc.from_monty_fast(pos[4])[active]
    assert half.sum() == 3 * (110 + 32) and hard.sum() == 3 * (110 + 1 + 32)   # chains + Merkle are half-output; + the pk-hash IV call
    b = ob.whir_builder(log_inv_rate=1, pow_bits=5, security=50)
    raw = ob.prove_execution(orc, w, synth_witness.header(w), b)          # raises "logup sum != 0" on any inconsistency
    ok, err = ob.verify_execution(orc, w, raw, b)
    assert ok, err
    lb = lm.WhirBuilder.default(1, security_level=50, pow_bits=5)
    cfg = lm.WhirConfig.new(lb, synth_witness.stacked_n_vars(w)).to_dict()
    sizes = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
    ok, err = lm.verify_execution(w, lm.Prover.from_raw(raw, sizes), lb)
    assert ok, err
    # a forged signature cannot be laid out: the last Merkle hash would have to overwrite the public key with another value
    bad = dict(w["xmss"], merkle_proof=w["xmss"]["merkle_proof"].copy())
    bad["merkle_proof"][1, 3, 2] ^= 1
    assert not Xmss(orc).verify(bad, w["memory"][112:120], SLOT)[1]


@pytest.mark.gpu
def test_device_proof_of_xmss_witness_equals_oracle(ctx, orc):
    w = xmss_witness.build(orc, np.random.default_rng(8), n_sigs=20, n_arith=120, slot=SLOT,
                           compress=lambda x: ctx.poseidon16(x, compress=True))
    assert w["log_rows"] == {0: 13, 1: 9, 2: 12}
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    lb = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    tr, keep = lm.make_execution_trace(ctx, w)
    pr = lm.Prover(ctx)
    pr.prove_execution(tr, lm.WhirConfig.new(lb, ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))))
    ref = ob.prove_execution(orc, w, synth_witness.header(w), b)
    assert np.array_equal(pr.proof(), ref)
    ok, err = lm.verify_execution(w, pr.proof_bytes(), lb)
    assert ok, err
