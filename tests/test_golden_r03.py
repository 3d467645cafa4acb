"""The second pin (SURVEY.md §8(c) "fixtures to commit"): tests/golden/vectors_r03.json was produced in the
build container by tests/golden/twin_r03.py — a pure-integer Python restatement written from the reference's Rust (textbook
Poseidon checked against the reference KAT, the evals-DFT by the identity of the reference's own test, sponge, Merkle tree,
challenger, PoW, GKR at log_n = 11, product sumcheck).  BOTH the C++ oracle (CPU tests) and the device / host library
(-m gpu tests) must reproduce every word of it."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob

V = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_r03.json")))
u32 = lambda x: np.asarray(x, dtype=np.uint32)  # noqa: E731
vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp)


# ---------------------------------------------------------------------------------------------------------------- oracle (CPU)
@pytest.mark.parametrize("name", ["commit_zero_suffix", "commit_dense"])
def test_oracle_commit_matches_twin(orc, name):
    v = V[name]
    mat = orc.lde_base(u32(v["evals"]), v["folding_factor"], v["log_inv_rate"], v["dft_n_cols"])
    assert np.array_equal(mat, u32(v["matrix"]))
    digests = orc.merkle_build(mat, 1 << v["folding_factor"])
    h = mat.shape[0]
    assert np.array_equal(digests[:h], u32(v["leaf_digests"])) and np.array_equal(digests[-1], u32(v["root"]))
    layers, off, n = [], 0, h
    while n >= 1:
        layers.append(digests[off:off + n])
        off += n
        n //= 2
    for idx, sib in zip(v["open_indices"], v["siblings"]):
        got = [layers[l][(idx >> l) ^ 1] for l in range(len(layers) - 1)]
        assert np.array_equal(np.stack(got), u32(sib))


def test_oracle_sponge_matches_twin(orc):
    v = V["sponge"]
    row = u32(v["row"])[None, :]
    assert np.array_equal(orc.merkle_build(np.concatenate([row, row]), v["full_width"])[0], u32(v["digest_plain"]))
    # 64-word leaf with 21 non-zero words: five all-zero rate chunks -> the precomputed zero-suffix state
    assert np.array_equal(orc.merkle_build(np.concatenate([row, row]), v["zero_suffix_full_width"])[0], u32(v["digest_zero_suffix"]))


def _oracle_ps(orc):
    lib = orc.lib
    lib.orc_ps_new.restype = vp
    for n in ("orc_ps_free", "orc_ps_add_base", "orc_ps_duplex", "orc_ps_sample_vec", "orc_ps_sample_in_range", "orc_ps_add_sumcheck_polynomial",
              "orc_ps_pow_grinding", "orc_ps_state"):
        getattr(lib, n).restype = None
    lib.orc_ps_transcript.restype = C.c_uint64
    return lib


def test_oracle_transcript_primitives_match_twin(orc):
    lib = _oracle_ps(orc)
    v = V["transcript"]
    coeffs, alpha = u32(v["poly_coeffs"]), u32(v["poly_eq_alpha"])
    h = vp(lib.orc_ps_new())
    first = u32(v["transcript"])[:11].copy()
    lib.orc_ps_add_base(h, _p(first), C.c_uint64(11))
    out = np.empty((3, 5), dtype=np.uint32)
    lib.orc_ps_sample_vec(h, C.c_uint64(3), _p(out))
    assert np.array_equal(out, u32(v["sample_vec_3"]))
    lib.orc_ps_duplex(h)
    q = np.empty(13, dtype=np.uint64)
    lib.orc_ps_sample_in_range(h, C.c_uint32(9), C.c_uint64(13), _p(q))
    assert list(q) == v["sample_in_range_9x13"]
    lib.orc_ps_add_sumcheck_polynomial(h, _p(coeffs), C.c_uint32(3), _p(alpha))
    s = np.empty((1, 5), dtype=np.uint32)
    lib.orc_ps_sample_vec(h, C.c_uint64(1), _p(s))
    assert np.array_equal(s[0], u32(v["sample_after_sumcheck_poly"]))
    st = np.empty(16, dtype=np.uint32)
    lib.orc_ps_state(h, _p(st))
    assert np.array_equal(st[:8], u32(v["pow_capacity"]))
    lib.orc_ps_pow_grinding(h, C.c_uint32(v["pow_bits"]))
    lib.orc_ps_duplex(h)
    lib.orc_ps_sample_vec(h, C.c_uint64(1), _p(s))
    assert np.array_equal(s[0], u32(v["sample_after_pow"]))
    n = lib.orc_ps_transcript(h, None)
    tr = np.empty(n, dtype=np.uint32)
    lib.orc_ps_transcript(h, _p(tr))
    assert np.array_equal(tr, u32(v["transcript"])) and tr[-1] == v["pow_witness"]
    lib.orc_ps_state(h, _p(st))
    assert np.array_equal(st, u32(v["final_state"]))
    lib.orc_ps_free(h)


def test_oracle_gkr_matches_twin(orc):
    v = V["gkr"]
    proof, q, pt, cl = ob.gkr_prove(orc, u32(v["nums"]), u32(v["dens"]))
    assert np.array_equal(q, u32(v["quotient"])) and np.array_equal(pt, u32(v["point"]))
    assert np.array_equal(cl[0], u32(v["claim_num"])) and np.array_equal(cl[1], u32(v["claim_den"]))
    t = u32(v["transcript"])
    assert proof[0] == t.size and np.array_equal(proof[1:1 + t.size], t)


def test_oracle_product_sumcheck_matches_twin(orc):
    v = V["product_sumcheck"]
    f, W, ch = u32(v["f"]), u32(v["W"]), u32(v["challenges"])
    n_vars, n_rounds = int(np.log2(f.size)), ch.shape[0]
    out = np.empty((n_rounds, 3, 5), dtype=np.uint32)
    fo = np.empty((f.size >> n_rounds, 5), dtype=np.uint32)
    wo = np.empty_like(fo)
    orc.lib.orc_product_sumcheck_fixed(_p(f), _p(W), C.c_uint32(n_vars), _p(ch), C.c_uint32(n_rounds), _p(out), _p(fo), _p(wo))
    assert np.array_equal(out, u32(v["rounds"])) and np.array_equal(fo, u32(v["f_final"])) and np.array_equal(wo, u32(v["W_final"]))


# ---------------------------------------------------------------------------------------------------------------- library (GPU / host)
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["commit_zero_suffix", "commit_dense"])
def test_device_commit_matches_twin(ctx, name):
    v = V[name]
    tree = ctx.commit(ctx.to_device(u32(v["evals"])), False, 10, v["folding_factor"], v["log_inv_rate"], actual_len=v["actual_len"])
    assert np.array_equal(tree.root, u32(v["root"]))
    h, full = len(v["matrix"]), 1 << v["folding_factor"]
    mat = tree.matrix()
    assert mat.shape == (h, full) and np.array_equal(mat[:, :v["dft_n_cols"]], u32(v["matrix"])) and not mat[:, v["dft_n_cols"]:].any()
    assert np.array_equal(tree.digests()[:h], u32(v["leaf_digests"]))
    leaves, sib = tree.open(v["open_indices"])
    for k, idx in enumerate(v["open_indices"]):
        assert np.array_equal(leaves[k][:v["dft_n_cols"]], u32(v["matrix"])[idx]) and np.array_equal(sib[k], u32(v["siblings"][k]))
    tree.free()


@pytest.mark.gpu
def test_library_transcript_and_pow_match_twin(ctx):
    v = V["transcript"]
    pr = lm.Prover(ctx)
    lib, h = pr.lib, pr.h
    first = u32(v["transcript"])[:11].copy()
    lib.lmh_add_base_scalars(h, _p(first), 11)
    out = np.empty((3, 5), dtype=np.uint32)
    assert lib.lmh_sample_vec(h, 3, _p(out)) == 0 and np.array_equal(out, u32(v["sample_vec_3"]))
    lib.lmh_duplex(h)
    q = np.empty(13, dtype=np.uint64)
    assert lib.lmh_sample_in_range(h, 9, 13, _p(q)) == 0 and list(q) == v["sample_in_range_9x13"]
    coeffs, alpha = u32(v["poly_coeffs"]), u32(v["poly_eq_alpha"])
    lib.lmh_add_sumcheck_polynomial(h, _p(coeffs), 3, _p(alpha))
    s = np.empty((1, 5), dtype=np.uint32)
    assert lib.lmh_sample_vec(h, 1, _p(s)) == 0 and np.array_equal(s[0], u32(v["sample_after_sumcheck_poly"]))
    assert np.array_equal(pr.state()[:8], u32(v["pow_capacity"]))
    assert lib.lmh_pow_grinding(ctx.h, h, v["pow_bits"]) == 0       # the search runs on the device: smallest witness
    lib.lmh_duplex(h)
    assert lib.lmh_sample_vec(h, 1, _p(s)) == 0 and np.array_equal(s[0], u32(v["sample_after_pow"]))
    blob = pr.proof()
    t = u32(v["transcript"])
    assert blob[0] == t.size and np.array_equal(blob[1:1 + t.size], t) and np.array_equal(pr.state(), u32(v["final_state"]))


@pytest.mark.gpu
def test_device_gkr_matches_twin(ctx):
    v = V["gkr"]
    pr = lm.Prover(ctx)
    q, pt, cl = pr.prove_gkr_quotient(ctx.to_device(u32(v["nums"])), ctx.ef_to_device_soa(u32(v["dens"])), v["log_n"])
    assert np.array_equal(q, u32(v["quotient"])) and np.array_equal(pt, u32(v["point"]))
    assert np.array_equal(cl[0], u32(v["claim_num"])) and np.array_equal(cl[1], u32(v["claim_den"]))
    blob, t = pr.proof(), u32(v["transcript"])
    assert blob[0] == t.size and np.array_equal(blob[1:1 + t.size], t)


@pytest.mark.gpu
def test_device_product_sumcheck_matches_twin(ctx):
    """lm_prod_round / lm_fold_round (one round per pass) and lm_prod_round2 / lm_fold2_round (two rounds per pass) against the
    twin's rounds with the same fixed challenges"""
    v = V["product_sumcheck"]
    f, W, ch, rounds = u32(v["f"]), u32(v["W"]), u32(v["challenges"]), u32(v["rounds"])
    n_vars = 8
    P = 0x7F000001
    add = lambda a, b: ((a.astype(np.uint64) + b) % P).astype(np.uint32)  # noqa: E731
    df, dW = ctx.to_device(f), ctx.ef_to_device_soa(W)
    c0, c2 = ctx.prod_round(df, False, dW, n_vars)
    assert np.array_equal(c0, rounds[0][0]) and np.array_equal(c2, rounds[0][2])
    f_ext = False
    for r in range(1, ch.shape[0]):
        df, dW, c0, c2 = ctx.fold_round(df, f_ext, dW, n_vars - r + 1, ch[r - 1])
        f_ext = True
        assert np.array_equal(c0, rounds[r][0]) and np.array_equal(c2, rounds[r][2]), r
    df2 = ctx.fold(df, True, n_vars - ch.shape[0] + 1, ch[-1])
    dW2 = ctx.fold(dW, True, n_vars - ch.shape[0] + 1, ch[-1])
    soa = lambda buf: buf.download().reshape(5, -1).T  # noqa: E731 — device SoA EF -> (n, 5)
    assert np.array_equal(soa(df2), u32(v["f_final"])) and np.array_equal(soa(dW2), u32(v["W_final"]))
