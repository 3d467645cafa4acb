"""GPU: degenerate and boundary inputs of the C ABI (the reference's tests exercise the same corners through tiny
instances: empty slices, single elements, fully padded tables)."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests.oracle_binding import P, rand_field

pytestmark = pytest.mark.gpu


def test_mle_eval_zero_and_one_variable(ctx, orc):
    rng = np.random.default_rng(1)
    v = rand_field(rng, 2)
    pt = rand_field(rng, (1, 5))
    got = ctx.mle_eval(ctx.to_device(v), False, 1, pt)[0]
    assert np.array_equal(got, orc.mle_eval_base(v, pt))
    one = rand_field(rng, 1)
    got0 = ctx.mle_eval(ctx.to_device(one), False, 0, np.zeros((0, 5), dtype=np.uint32))[0]
    assert list(got0) == [int(one[0]), 0, 0, 0, 0]


def test_commit_single_nonzero_column_and_single_row(ctx, orc):
    """actual_len covering one column only (every other leaf word comes from the zero-suffix state), and the smallest
    tree the WHIR schedule can ask for."""
    rng = np.random.default_rng(2)
    n_vars, fold, rate = 10, 4, 1
    poly = np.zeros(1 << n_vars, dtype=np.uint32)
    actual = 1 << (n_vars - fold)
    poly[:actual] = rand_field(rng, actual)
    tree = ctx.commit(ctx.to_device(poly), False, n_vars, fold, rate, actual_len=actual)
    layers = orc.merkle_build(orc.lde_base(poly, fold, rate), 1 << fold)
    assert np.array_equal(tree.digests(), layers)
    # h = 2: n_vars + rate - fold = 1
    poly2 = rand_field(rng, 16)
    t2 = ctx.commit(ctx.to_device(poly2), False, 4, 4, 1)
    assert np.array_equal(t2.digests(), orc.merkle_build(orc.lde_base(poly2, 4, 1), 16))


def test_pow_grind_trivial_bits(ctx):
    cap = rand_field(np.random.default_rng(3), 8)
    assert ctx.pow_grind(cap, 0) == 0
    w1 = ctx.pow_grind(cap, 1)
    assert 0 <= w1 < P


def test_poseidon_batch_empty_is_a_no_op(ctx):
    assert ctx.poseidon16(np.zeros((0, 16), dtype=np.uint32)).shape == (0, 16)


def test_gkr_smallest_instance_all_padding(ctx, orc):
    """n = 6 (one layer above the 2^5 values sent in clear), every entry the neutral fraction 0/1."""
    n = 6
    nums = np.zeros(1 << n, dtype=np.uint32)
    dens = np.zeros((1 << n, 5), dtype=np.uint32)
    dens[:, 0] = int(orc.to_monty(1))
    ref = ob.gkr_prove(orc, nums, dens)
    pr = lm.Prover(ctx)
    dn = ctx.to_device(nums)
    dd = ctx.ef_to_device_soa(dens)
    q, pt, cl = pr.prove_gkr_quotient(dn, dd, n)
    assert np.array_equal(pr.proof(), ref[0]) and list(q) == [0, 0, 0, 0, 0]
