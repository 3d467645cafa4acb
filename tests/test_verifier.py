"""The library's verifier (lmh_verify_execution), CPU: it accepts the ORACLE prover's proofs — two
independent restatements of the reference's prover / verifier pair meeting in the middle — through every entry form (prover
object, decoded proof, postcard bytes, lz4 frame), agrees with the oracle's verifier on tampered proofs, and rejects a wrong
instance (public input, bytecode, bytecode hash, ending pc, security parameters)."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests import synth_witness


def _batch_sizes(w, builder):
    cfg = lm.WhirConfig.new(builder, synth_witness.stacked_n_vars(w)).to_dict()
    return [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]


@pytest.fixture(scope="module")
def small(orc):
    w = synth_witness.build(orc, np.random.default_rng(31), n_calls=40)
    ob_b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    lm_b = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    raw = ob.prove_execution(orc, w, synth_witness.header(w), ob_b)
    return w, raw, ob_b, lm_b


def test_accepts_oracle_proof_in_every_form(orc, small):
    w, raw, ob_b, lm_b = small
    pr = lm.Prover.from_raw(raw, _batch_sizes(w, lm_b))
    ok, err = lm.verify_execution(w, pr, lm_b)
    assert ok, err
    data, comp = pr.proof_bytes(), pr.proof_bytes(compressed=True)
    for proof, kw in ((data, {}), (comp, dict(compressed=True)), (lm.DecodedProof(data), {})):
        ok, err = lm.verify_execution(w, proof, lm_b, **kw)
        assert ok, err
    # default builder (124-bit) on a 60-bit proof: the schedule differs, so the transcript cannot line up
    ok, err = lm.verify_execution(w, data)
    assert not ok and err


def test_mixed_program_and_other_rate(orc):
    w = synth_witness.build_mixed(orc, np.random.default_rng(32))
    w["log_inv_rate"] = 2
    ob_b = ob.whir_builder(log_inv_rate=2, pow_bits=5, security=50)
    lm_b = lm.WhirBuilder.default(2, security_level=50, pow_bits=5)
    raw = ob.prove_execution(orc, w, synth_witness.header(w), ob_b)
    ok, err = lm.verify_execution(w, lm.Prover.from_raw(raw, _batch_sizes(w, lm_b)), lm_b)
    assert ok, err


def test_rejects_wrong_instance(orc, small):
    w, raw, ob_b, lm_b = small
    pr = lm.Prover.from_raw(raw, _batch_sizes(w, lm_b))
    for key, mutate in (("public_input", lambda a: np.concatenate([a[:3], [(int(a[3]) + 1) % 0x7F000001], a[4:]]).astype(np.uint32)),
                        ("public_input", lambda a: a[:-1]),
                        ("bytecode_hash", lambda a: np.roll(a, 1)),
                        ("ending_pc", lambda v: v + 1)):
        w2 = dict(w)
        w2[key] = mutate(w[key])
        ok, err = lm.verify_execution(w2, pr, lm_b)
        assert not ok and err, key
    w2 = dict(w, bytecode=w["bytecode"].copy())
    w2["bytecode"][3, 1] ^= 1  # an instruction the program executes
    ok, err = lm.verify_execution(w2, pr, lm_b)
    assert not ok and "logup" in err
    ok, err = lm.verify_execution(w, pr, lm.WhirBuilder.default(1, security_level=60, pow_bits=7))
    assert not ok


def test_tampering_matches_the_oracle_verifier(orc, small):
    """single-word corruptions all over the proof: both verifiers must reject every one of them"""
    w, raw, ob_b, lm_b = small
    sizes = _batch_sizes(w, lm_b)
    rng = np.random.default_rng(33)
    T = int(raw[0])
    positions = list(rng.integers(1, 1 + T, size=25)) + list(rng.integers(2 + T, raw.size, size=25))
    rejected = 0
    for pos in positions:
        bad = raw.copy()
        bad[pos] = (int(bad[pos]) + 1 + int(rng.integers(0, 1000))) % 0x7F000001
        try:
            pr = lm.Prover.from_raw(bad, sizes)
        except lm.LmError:
            rejected += 1  # a structural word (length / index) no longer parses
            continue
        ok_lm, _ = lm.verify_execution(w, pr, lm_b)
        try:
            # what travels is the pruned proof: a corrupted sibling that pruning drops (the verifier recomputes it from the
            # neighbouring path) is not part of it, so the oracle checks the same restored proof
            ok_orc, _ = ob.verify_execution(orc, w, ob.restore_proof(orc, ob.prune_proof(orc, bad, sizes)), ob_b)
        except Exception:  # noqa: BLE001 — the oracle binding raises on unparsable blobs
            ok_orc = False
        assert ok_lm == ok_orc, pos
        rejected += not ok_lm
    assert rejected >= 30  # (redundant siblings are pruned away before the proof travels: corrupting one changes nothing)


def _mutations(rng, raw, n):
    """n mutated copies of a raw proof blob: word edits of several kinds, swaps, block moves, truncations and extensions"""
    P = 0x7F000001
    T = int(raw[0])
    for k in range(n):
        bad = raw.copy()
        kind = k % 8
        pos = int(rng.integers(1, raw.size))
        if kind == 0:
            bad[pos] = (int(bad[pos]) + 1 + int(rng.integers(0, P - 2))) % P
        elif kind == 1:
            bad[pos] = 0 if bad[pos] else 1
        elif kind == 2:
            bad[pos] = P - 1 if int(bad[pos]) != P - 1 else P - 2
        elif kind == 3:
            q = int(rng.integers(1, raw.size))
            if bad[pos] == bad[q]:
                bad[pos] = (int(bad[pos]) + 1) % P
            else:
                bad[pos], bad[q] = bad[q], bad[pos]
        elif kind == 4:  # a block of the transcript moved by one word
            a = int(rng.integers(1, max(2, T - 8)))
            bad[a:a + 8] = np.roll(bad[a:a + 8], 1)
            if np.array_equal(bad, raw):
                bad[a] = (int(bad[a]) + 1) % P
        elif kind == 5:  # truncated
            bad = bad[:raw.size - int(rng.integers(1, 16))]
        elif kind == 6:  # extended with field words
            bad = np.concatenate([bad, ob.rand_field(rng, int(rng.integers(1, 9)))]).astype(np.uint32)
        else:            # the transcript length word itself
            bad[0] = max(1, T + int(rng.integers(-3, 4)) or T + 1)
        yield k, bad


@pytest.mark.parametrize("which", ["small", "mixed"])
def test_verifier_differential_fuzz_against_the_oracle_verifier(orc, small, which):
    """lmh_verify_execution is load-bearing beyond accepting proofs (it rebuilds the raw transcripts and opening claims the recursion
    program is fed), and only tests/test_verifier.py::test_tampering... compared it with anything: 50 single-word edits.  Here 2 x 240
    mutations of eight kinds — field edits to small / large / swapped values, a rotated transcript block, truncation, extension, a changed
    length word — on two instances (two rates, two programs): whatever parses must get the SAME verdict from the library's verifier and
    from the oracle's (on the proof as it travels: pruned and restored), and no mutation may be accepted."""
    if which == "small":
        w, raw, ob_b, lm_b = small
    else:
        w = synth_witness.build_mixed(orc, np.random.default_rng(35))
        w["log_inv_rate"] = 2
        ob_b = ob.whir_builder(log_inv_rate=2, pow_bits=5, security=50)
        lm_b = lm.WhirBuilder.default(2, security_level=50, pow_bits=5)
        raw = ob.prove_execution(orc, w, synth_witness.header(w), ob_b)
    sizes = _batch_sizes(w, lm_b)
    ok, err = lm.verify_execution(w, lm.Prover.from_raw(raw, sizes), lm_b)
    assert ok, err
    rng = np.random.default_rng(36)
    compared = unparsable = 0
    for k, bad in _mutations(rng, raw, 240):
        try:
            pr = lm.Prover.from_raw(bad, sizes)
        except lm.LmError:
            unparsable += 1
            continue
        ok_lm, _ = lm.verify_execution(w, pr, lm_b)
        try:
            travelled = ob.restore_proof(orc, ob.prune_proof(orc, bad, sizes))
            ok_orc, _ = ob.verify_execution(orc, w, travelled, ob_b)
            same_as_original = travelled.size == raw.size and np.array_equal(travelled, raw)
        except Exception:  # noqa: BLE001 — the oracle binding raises on unparsable blobs
            ok_orc, same_as_original = False, False
        assert ok_lm == ok_orc, (k, k % 8)
        assert not ok_lm or same_as_original, (k, k % 8)   # accepted only when the edit fell on a sibling that pruning drops
        compared += 1
    assert compared >= 120, (compared, unparsable)


def test_external_pin_fixture_of_real_signatures_is_accepted_by_the_library_verifier():
    """tests/golden/external_pin_xmss (tools/write_proof.py --xmss: a device proof of the aggregation program on 40 real XMSS signatures,
    VM run with the parallel batch on the device, default_whir_config) — what `make pin` feeds to the REFERENCE's verify_execution —
    is accepted by lmh_verify_execution, and a flipped public-input word is rejected.  Host code only."""
    import os
    import zlib
    from leanmultisig_amd.vm import to_monty
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "external_pin_xmss")
    c = np.frombuffer(zlib.decompress(open(os.path.join(d, "instance.zlib"), "rb").read()), dtype="<u4")
    lb, ep, n_pub = int(c[0]), int(c[1]), int(c[2])
    rows = np.zeros((1 << lb, 16), dtype=np.uint32)
    rows[:, :12] = to_monty(c[11 + n_pub:].reshape(-1, 12)).astype(np.uint32)
    w = dict(log_bytecode=lb, ending_pc=ep, public_input=to_monty(c[11:11 + n_pub]).astype(np.uint32), bytecode_hash=to_monty(c[3:11]).astype(np.uint32),
             bytecode=rows)
    proof = open(os.path.join(d, "proof.bin"), "rb").read()
    ok, err = lm.verify_execution(w, proof, None)
    assert ok, err
    bad = dict(w, public_input=np.roll(w["public_input"], 1))
    assert not lm.verify_execution(bad, proof, None)[0]
