"""GPU parity: WHIR commitment (LDE + Merkle) and MLE evaluation through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests.oracle_binding import P, rand_field

pytestmark = pytest.mark.gpu

KAT_OUT = [610090613, 935319874, 1893335292, 796792199, 356405232, 552237741, 55134556, 1215104204, 1823723405,
           1133298033, 1780633798, 1453946561, 710069176, 1128629550, 1917333254, 1175481618]


def test_poseidon_kat_on_device(ctx, orc):
    st = orc.to_monty(np.arange(16))
    out = ctx.poseidon16(st)[0]
    assert list(orc.from_monty(out)) == KAT_OUT


def test_poseidon_batch_matches_oracle(ctx, orc):
    rng = np.random.default_rng(0)
    st = rand_field(rng, (1000, 16))
    st[0] = P - 1
    st[1] = 0
    assert np.array_equal(ctx.poseidon16(st), orc.poseidon16_permute(st))
    assert np.array_equal(ctx.poseidon16(st, compress=True), orc.poseidon16_compress(st))


@pytest.mark.parametrize("n", [1, 3, 64, 1000, 5000])
def test_poseidon_quad_matches_oracle(ctx, orc, n):
    """the 4-lane cooperative permutation (poseidon16_quad.h, used by the proof-of-work search): every word of every state"""
    rng = np.random.default_rng(100 + n)
    st = rand_field(rng, (n, 16))
    st[0] = P - 1
    if n > 1:
        st[1] = 0
    if n > 2:
        st[2] = orc.to_monty(np.arange(16))
    assert np.array_equal(ctx.poseidon16(st, quad=True), orc.poseidon16_permute(st))
    assert np.array_equal(ctx.poseidon16(st, compress=True, quad=True), orc.poseidon16_compress(st))


@pytest.mark.parametrize("n_vars,fold,rate,frac", [
    (7, 4, 1, 1.0),      # tiny: h = 16 rows, 16 columns
    (10, 7, 1, 1.0),     # 128 columns, h = 16
    (12, 7, 2, 1.0),
    (14, 7, 1, 0.79),    # zero tail -> zero-suffix sponge state (>= 2 zero chunks)
    (14, 7, 1, 0.93),    # zero tail of exactly one chunk -> plain path
    (16, 7, 1, 0.40),
    (18, 4, 3, 1.0),     # h = 2^17 rows: 12 + 5 layer passes
    (19, 5, 0, 1.0),     # rate 1, h = 2^14
    (23, 4, 1, 1.0),     # h = 2^20 rows: the register-resident radix-16 passes (12 layers + 8 layers), rate 1/2 (uint4 loads, layer 1 skipped)
    (22, 4, 2, 1.0),     # the same at rate 1/4 (layers 1-2 skipped)
    (21, 4, 3, 0.9),     # rate 1/8, a zero tail
])
def test_commit_base_matches_oracle(ctx, orc, n_vars, fold, rate, frac):
    rng = np.random.default_rng(n_vars * 100 + fold)
    n = 1 << n_vars
    actual = max(1, int(n * frac))
    evals = rand_field(rng, n)
    evals[actual:] = 0
    tree = ctx.commit(ctx.to_device(evals), False, n_vars, fold, rate, actual_len=actual)
    ref_rows = orc.lde_base(evals, fold, rate)
    assert tree.log_height == n_vars + rate - fold
    assert np.array_equal(tree.matrix(), ref_rows)
    ref_layers = orc.merkle_build(ref_rows, 1 << fold)
    assert np.array_equal(tree.digests(), ref_layers)
    assert np.array_equal(tree.root, ref_layers[-1])
    h = 1 << tree.log_height
    idx = np.unique(np.concatenate([[0, h - 1], rng.integers(0, h, size=9)]))
    leaves, sib = tree.open(idx)
    for k, i in enumerate(idx):
        assert np.array_equal(leaves[k], ref_rows[int(i)])
        assert orc.merkle_verify(tree.root, tree.log_height, int(i), leaves[k], sib[k])


@pytest.mark.parametrize("n_vars,fold,rate", [(8, 5, 1), (12, 5, 2), (15, 5, 1), (17, 4, 3)])
def test_commit_ext_matches_oracle(ctx, orc, n_vars, fold, rate):
    rng = np.random.default_rng(n_vars)
    evals = rand_field(rng, (1 << n_vars, 5))
    tree = ctx.commit(ctx.ef_to_device_soa(evals), True, n_vars, fold, rate)
    ref_rows = orc.lde_ext(evals, fold, rate)
    assert tree.leaf_words == 5 << fold
    assert np.array_equal(tree.matrix(), ref_rows)
    ref_layers = orc.merkle_build(ref_rows, 5 << fold)
    assert np.array_equal(tree.digests(), ref_layers)
    h = 1 << tree.log_height
    idx = rng.integers(0, h, size=7)
    leaves, sib = tree.open(idx)
    for k, i in enumerate(idx):
        assert np.array_equal(leaves[k], ref_rows[int(i)])
        assert orc.merkle_verify(tree.root, tree.log_height, int(i), leaves[k], sib[k])


@pytest.mark.parametrize("n_vars", [0, 1, 5, 12, 13, 17])
def test_mle_eval_matches_oracle(ctx, orc, n_vars):
    rng = np.random.default_rng(n_vars)
    n = 1 << n_vars
    pt = rand_field(rng, (n_vars, 5))
    polys = rand_field(rng, (3, n))
    got = ctx.mle_eval(ctx.to_device(polys), False, n_vars, pt, n_polys=3, stride_words=n)
    for k in range(3):
        assert list(got[k]) == list(orc.mle_eval_base(polys[k], pt))
    ev = rand_field(rng, (n, 5))
    got = ctx.mle_eval(ctx.ef_to_device_soa(ev), True, n_vars, pt)
    assert list(got[0]) == list(orc.mle_eval_ext(ev, pt))


@pytest.mark.parametrize("n_vars,n_points", [(0, 2), (3, 1), (11, 2), (13, 3), (18, 2), (16, 5)])
def test_mle_eval_points_matches_oracle(ctx, orc, n_vars, n_points):
    """lm_mle_eval_points (the OOD samples of a commitment in one pass per pair of points): every point against the oracle, base and
    extension-field polynomials; and the same evaluations through the deferred-result window (lm_results_defer_begin / _end)"""
    rng = np.random.default_rng(1000 + n_vars * 10 + n_points)
    n = 1 << n_vars
    pts = rand_field(rng, (n_points, n_vars, 5))
    poly = rand_field(rng, n)
    d = ctx.to_device(poly)
    got = ctx.mle_eval_points(d, False, n_vars, pts)
    for q in range(n_points):
        assert list(got[q]) == list(orc.mle_eval_base(poly, pts[q]))
    ev = rand_field(rng, (n, 5))
    de = ctx.ef_to_device_soa(ev)
    got = ctx.mle_eval_points(de, True, n_vars, pts)
    for q in range(n_points):
        assert list(got[q]) == list(orc.mle_eval_ext(ev, pts[q]))
    jobs = [(d, False, n_vars, pts[q]) for q in range(n_points)] + [(de, True, n_vars, pts[q]) for q in range(n_points)]
    res = ctx.mle_eval_deferred(jobs)
    for q in range(n_points):
        assert list(res[q]) == list(orc.mle_eval_base(poly, pts[q]))
        assert list(res[n_points + q]) == list(orc.mle_eval_ext(ev, pts[q]))


def test_tree_open_in_two_calls(ctx, orc):
    """lm_tree_open_begin / _end with other work enqueued in between (WHIR enqueues the round's weight kernels there) == lm_tree_open"""
    rng = np.random.default_rng(77)
    n_vars, fold, rate = 16, 7, 1
    evals = rand_field(rng, 1 << n_vars)
    d = ctx.to_device(evals)
    tree = ctx.commit(d, False, n_vars, fold, rate)
    h = 1 << tree.log_height
    idx = rng.integers(0, h, size=40)
    end = tree.open_begin(idx)
    pt = rand_field(rng, (n_vars, 5))
    val = ctx.mle_eval(d, False, n_vars, pt)  # uses the context's scratch and publishes a later sequence number
    leaves, sib = end()
    ref_leaves, ref_sib = tree.open(idx)
    assert np.array_equal(leaves, ref_leaves) and np.array_equal(sib, ref_sib)
    assert list(val[0]) == list(orc.mle_eval_base(evals, pt))
    for k, i in enumerate(idx):
        assert orc.merkle_verify(tree.root, tree.log_height, int(i), leaves[k], sib[k])


def test_ood_point_is_dft_consistent(ctx, orc):
    """size-independent property at a larger size: LDE row i of a column == MLE of the (replicated) column at
    expand_from_univariate(g^i) (whir/src/dft.rs:583-603), evaluated entirely on the device."""
    rng = np.random.default_rng(7)
    n_vars, fold, rate = 22, 7, 1
    evals = rand_field(rng, 1 << n_vars)
    d = ctx.to_device(evals)
    tree = ctx.commit(d, False, n_vars, fold, rate)
    log_h = tree.log_height
    h = 1 << log_h
    col_len = 1 << (n_vars - fold)
    g = orc.lib.orc_two_adic_generator(log_h)
    one = int(orc.to_monty(1))
    idx = [0, 1, 12345, h - 1]
    leaves, sib = tree.open(idx)
    for k, i in enumerate(idx):
        assert orc.merkle_verify(tree.root, log_h, i, leaves[k], sib[k])
        gi = pow(int(orc.lib.orc_from_monty(g)), i, P)
        pt = orc.expand_from_univariate(np.array([orc.to_monty(gi), 0, 0, 0, 0], dtype=np.uint32), log_h)
        # replicated column in log_h variables: last `rate` variables are dummies -> drop them from the point
        pt_col = pt[: log_h - rate]
        for c in (0, 77, 127):
            got = ctx.mle_eval(d.ptr + 4 * c * col_len, False, n_vars - fold, pt_col)
            assert got[0][0] == leaves[k][c] and not got[0][1:].any()


@pytest.mark.parametrize("aligned", [True, False])
def test_stack_columns(ctx, aligned):
    """stack_polynomials (stacked_pcs.rs:99-157): zero-padded concatenation in one pass; unaligned pieces take the copy path."""
    rng = np.random.default_rng(3)
    total = 1 << 14
    src = rng.integers(0, 0x7F000001, size=1 << 13, dtype=np.uint32)
    d_src = ctx.to_device(src)
    if aligned:
        jobs = [(0, 0, 1024), (1024, 1024, 256), (1280, 2048, 4), (4096, 4096, 4096), (100 * 4, 12288, 1000)]
    else:
        jobs = [(0, 0, 1023), (1024, 1025, 256), (4096, 4096, 4095), (3, 9000, 7)]
    out = ctx.stack_columns(total, [(d_src, so, do, n) for so, do, n in jobs]).download()
    want = np.zeros(total, dtype=np.uint32)
    for so, do, n in jobs:
        want[do:do + n] = src[so:so + n]
    assert np.array_equal(out, want)
    assert np.array_equal(ctx.stack_columns(64, []).download(), np.zeros(64, dtype=np.uint32))


@pytest.mark.parametrize("n", [1, 3, 17, 4096, 16384, 16385, 40000])
def test_poseidon_batch_both_kernels(ctx, orc, n):
    """lm_poseidon16_permute / compress switch to the 16-lane cooperative kernel at small n (poseidon16_coop.h): same
    permutation on both sides of the threshold, checked against the textbook oracle."""
    rng = np.random.default_rng(n)
    st = rand_field(rng, (n, 16))
    got_p = ctx.poseidon16(st)
    got_c = ctx.poseidon16(st, compress=True)
    idx = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, size=min(n, 40))]))
    for i in idx:
        want = orc.poseidon16_permute(st[i:i + 1])[0]
        assert np.array_equal(got_p[i], want), i
        assert np.array_equal(got_c[i], (want.astype(np.uint64) + st[i]) % P), i
