"""CPU tests of the oracle's WHIR restatement: parameter derivation and prove -> verify round trips with the shapes of
the reference's own protocol tests (whir/tests/run_whir.rs)."""
import numpy as np

from tests import oracle_binding as ob


def test_default_config_shapes(orc):
    # lean_prover defaults (lean_prover/src/lib.rs:22-50) at the config-2 size: n = 26, rate 1/2
    cfg = ob.whir_config(orc, ob.whir_builder(log_inv_rate=1), 26)
    assert cfg["n_rounds"] == 3 and cfg["final_sumcheck_rounds"] == 4
    assert [r["num_variables"] for r in cfg["rounds"]] == [19, 14, 9]
    assert [r["log_inv_rate"] for r in cfg["rounds"]] == [1, 3, 7]
    # query counts decrease with the rate; PoW never exceeds the budget
    q = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
    assert q == sorted(q, reverse=True)
    assert all(r["folding_pow_bits"] <= 16 and r["query_pow_bits"] <= 16 for r in cfg["rounds"])
    assert cfg["commitment_ood_samples"] >= 1
    print(cfg)


def _roundtrip(orc, n, rate, pow_bits, seed, actual_frac=1.0, small_folds=True):
    rng = np.random.default_rng(seed)
    if small_folds:  # fold 4 then 3, send coefficients at <= 3 variables: several WHIR rounds at tiny sizes
        b = ob.whir_builder(log_inv_rate=rate, pow_bits=pow_bits, security=60 + pow_bits, fold_first=4, fold_sub=3,
                            max_send=3, rs_red=3)
    else:
        b = ob.whir_builder(log_inv_rate=rate, pow_bits=pow_bits, security=60 + pow_bits)
    poly = ob.rand_field(rng, 1 << n)
    actual = int((1 << n) * actual_frac)
    poly[actual:] = 0
    sts = ob.random_statements(orc, rng, poly, n, n_points=4)
    proof, pt, perms = ob.whir_prove(orc, b, n, poly, sts, actual_len=actual, prefix=(1, 2, 3))
    ok, vpt, err = ob.whir_verify(orc, b, n, proof, sts, prefix=(1, 2, 3))
    assert ok, err
    assert np.array_equal(pt, vpt)
    return b, poly, sts, proof


def test_prove_verify_roundtrip(orc):
    _roundtrip(orc, 12, 1, 6, 0)
    _roundtrip(orc, 13, 2, 5, 1, actual_frac=0.7)
    _roundtrip(orc, 16, 1, 8, 3, small_folds=False)  # lean_prover fold schedule 7/5, one WHIR round


def test_verifier_rejects_tampering(orc):
    b, poly, sts, proof = _roundtrip(orc, 12, 1, 6, 2)
    bad = proof.copy()
    bad[5] ^= 1
    ok, _, _ = ob.whir_verify(orc, b, 12, bad, sts, prefix=(1, 2, 3))
    assert not ok
    # wrong claimed value
    sts2 = [dict(s) for s in sts]
    sel, v = sts2[0]["values"][0]
    v = v.copy()
    v[0] = (int(v[0]) + 1) % ob.P
    sts2[0]["values"] = [(sel, v)] + sts2[0]["values"][1:]
    ok, _, _ = ob.whir_verify(orc, b, 12, proof, sts2, prefix=(1, 2, 3))
    assert not ok
    # tampered merkle leaf (last word of the blob belongs to an opening path)
    bad = proof.copy()
    bad[-1] ^= 1
    ok, _, _ = ob.whir_verify(orc, b, 12, bad, sts, prefix=(1, 2, 3))
    assert not ok
