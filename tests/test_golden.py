"""Committed fixtures (tests/golden/): the reference's Poseidon1-16 KAT and oracle-generated vectors for every stage
(tests/golden/make_vectors.py).  CPU: the oracle still reproduces them (it cannot drift silently).  GPU: the device path
reproduces them WITHOUT consulting the oracle at run time."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import oracle_binding as ob
from tests import synth_witness

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(GOLD, "vectors_r01.json")))
KAT = json.load(open(os.path.join(GOLD, "poseidon1_16_kat.json")))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


def _commit_inputs():
    c = VEC["commit_base"]
    rng = np.random.default_rng(c["seed"])
    poly = ob.rand_field(rng, 1 << c["n_vars"])
    poly[c["actual_len"]:] = 0
    return c, poly


# ---- CPU: oracle vs fixtures -------------------------------------------------------------------------------------------
def test_oracle_reproduces_reference_kat(orc):
    out = orc.poseidon16_permute(orc.to_monty(np.array(KAT["input"])))[0]
    assert list(orc.from_monty(out)) == KAT["output"]


def test_oracle_reproduces_vectors(orc):
    from tests.golden import make_vectors
    fresh = make_vectors.vectors(orc)
    assert fresh == VEC


# ---- GPU: device vs fixtures ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_poseidon_matches_fixtures(ctx, orc):
    kat_in = orc.to_monty(np.array(KAT["input"])).reshape(1, 16)
    assert list(orc.from_monty(ctx.poseidon16(kat_in)[0])) == KAT["output"]
    st = np.array(VEC["poseidon16"]["states"], dtype=np.uint32)
    assert ctx.poseidon16(st).tolist() == VEC["poseidon16"]["permute"]
    assert ctx.poseidon16(st, compress=True).tolist() == VEC["poseidon16"]["compress"]
    big = np.tile(st, (5000, 1))  # the one-permutation-per-lane kernel (above the 16-lane threshold)
    assert ctx.poseidon16(big)[-4:].tolist() == VEC["poseidon16"]["permute"]


@pytest.mark.gpu
def test_device_commit_matches_fixtures(ctx):
    c, poly = _commit_inputs()
    tree = ctx.commit(ctx.to_device(poly), False, c["n_vars"], c["fold"], c["log_inv_rate"], actual_len=c["actual_len"])
    assert list(tree.root) == c["root"]
    assert digest(tree.digests()) == c["digests_sha256"]
    assert digest(tree.matrix()) == c["lde_sha256"]  # rows of the full leaf width (zero columns included)


@pytest.mark.gpu
def test_device_prove_execution_matches_fixture(ctx, orc):
    import leanmultisig_amd as lm
    e = VEC["prove_execution"]
    w = synth_witness.build(orc, np.random.default_rng(e["seed"]), n_calls=e["n_calls"])  # input generation only
    b = np.array(e["builder"], dtype=np.uint32)
    tr, keep = lm.make_execution_trace(ctx, w)
    # the WHIR schedule comes from the library's own WhirConfig::new: nothing of the oracle is consulted after input generation
    cfg = lm.WhirConfig.new(lm.WhirBuilder.default(int(b[0]), security_level=int(b[6]), pow_bits=int(b[7])),
                            ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr)))
    pr = lm.Prover(ctx)
    pr.prove_execution(tr, cfg)
    proof = pr.proof()
    assert proof.size == e["proof_words"] and digest(proof) == e["proof_sha256"]
    # ... and the reference's wire bytes (postcard of Proof<F>) of that proof, tests/golden/vectors_r02.json
    g = json.load(open(os.path.join(GOLD, "vectors_r02.json")))["proof_bytes"]
    data = pr.proof_bytes()
    assert list(pr.batch_sizes()) == g["batch_sizes"] and pr.proof_size_fe() == g["proof_size_fe"]
    assert len(data) == g["postcard_len"] and hashlib.sha256(data).hexdigest() == g["postcard_sha256"]
    assert np.array_equal(lm.DecodedProof(pr.proof_bytes(compressed=True), compressed=True).pruned_words(), pr.proof_pruned())
