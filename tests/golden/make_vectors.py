"""Generate tests/golden/vectors_r01.json: small input/output vectors for every stage of the path, produced by the CPU
oracle (oracle/, pinned by the reference's Poseidon KAT and identities — tests/test_oracle_pins.py).  The reference is a
Rust workspace and cannot be built in this image (no Rust toolchain), so no vector here comes from running it; the one
stored vector the reference holds for this path (the Poseidon1-16 KAT, poseidon1_koalabear_16.rs:1083-1091) is
tests/golden/poseidon1_16_kat.json.  Inputs are regenerated from the recorded seeds; outputs are stored in full when
small, as SHA-256 of the little-endian u32 words when large.
usage: python tests/golden/make_vectors.py   (from the repo root; needs oracle/liblm_oracle.so)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tests import oracle_binding as ob  # noqa: E402
from tests import synth_witness  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


def vectors(orc):
    v = {}
    rng = np.random.default_rng(2026)
    st = ob.rand_field(rng, (4, 16))
    v["poseidon16"] = {"seed": 2026, "states": st.tolist(), "permute": orc.poseidon16_permute(st).tolist(),
                       "compress": orc.poseidon16_compress(st).tolist()}
    # commitment: 2^10 base evaluations, fold 4, rate 1/2, 11 of 16 columns non-zero
    rng = np.random.default_rng(11)
    n_vars, fold, rate = 10, 4, 1
    poly = ob.rand_field(rng, 1 << n_vars)
    actual = 11 << (n_vars - fold)
    poly[actual:] = 0
    rows = orc.lde_base(poly, fold, rate)
    layers = orc.merkle_build(rows, 1 << fold)
    v["commit_base"] = {"seed": 11, "n_vars": n_vars, "fold": fold, "log_inv_rate": rate, "actual_len": actual,
                        "lde_sha256": digest(rows), "digests_sha256": digest(layers), "root": layers[-1].tolist()}
    # WHIR open (small parameters), GKR, execution slice: whole transcripts
    rng = np.random.default_rng(12)
    b = ob.whir_builder(log_inv_rate=1, max_send=3, rs_red=3, fold_first=4, fold_sub=3, pow_bits=4, security=40)
    n = 12
    poly = ob.rand_field(rng, 1 << n)
    sts = ob.random_statements(orc, rng, poly, n, n_points=3, with_next=True)
    proof = ob.whir_prove(orc, b, n, poly, sts)[0]
    v["whir"] = {"seed": 12, "n_vars": n, "builder": [int(x) for x in b], "proof_words": int(proof.size), "proof_sha256": digest(proof)}
    rng = np.random.default_rng(13)
    nums, dens = ob.gkr_instance(orc, rng, 9, 0.7)
    gp = ob.gkr_prove(orc, nums, dens)
    v["gkr"] = {"seed": 13, "log_n": 9, "active_frac": 0.7, "proof_sha256": digest(gp[0] if isinstance(gp, tuple) else gp)}
    rng = np.random.default_rng(14)
    w = synth_witness.build(orc, rng, n_calls=24)
    b2 = ob.whir_builder(log_inv_rate=1, pow_bits=5, security=50)
    pe = ob.prove_execution(orc, w, synth_witness.header(w), b2)
    v["prove_execution"] = {"seed": 14, "n_calls": 24, "builder": [int(x) for x in b2], "proof_words": int(pe.size),
                            "proof_sha256": digest(pe)}
    return v


if __name__ == "__main__":
    orc = ob.load()
    out = os.path.join(ROOT, "tests", "golden", "vectors_r01.json")
    json.dump(vectors(orc), open(out, "w"), indent=1)
    print("wrote", out)
