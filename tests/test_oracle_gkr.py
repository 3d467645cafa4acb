"""CPU test of the oracle's GKR restatement with the checks of the reference's own test
(quotient_gkr/mod.rs:262-279): quotient == sum n/d, final claims == MLE evaluations at the returned point."""
import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.oracle_binding import P


@pytest.mark.parametrize("log_n,frac", [(6, 1.0), (9, 0.6), (11, 0.51), (11, 1.0)])
def test_gkr_quotient_roundtrip(orc, log_n, frac):
    rng = np.random.default_rng(log_n)
    nums, dens = ob.gkr_instance(orc, rng, log_n, frac)
    proof, q, pt, cl = ob.gkr_prove(orc, nums, dens)
    ok, vq, vpt, vcl, err = ob.gkr_verify(orc, proof, log_n)
    assert ok, err
    assert np.array_equal(q, vq) and np.array_equal(pt, vpt) and np.array_equal(cl, vcl)
    # real quotient
    acc = np.zeros(5, dtype=np.uint64)
    for n_i, d_i in zip(nums, dens):
        t = orc.ef_mul(np.array([n_i, 0, 0, 0, 0], dtype=np.uint32), orc.ef_inv(d_i))
        acc = (acc + t) % P
    assert list(acc.astype(np.uint32)) == list(q)
    assert list(orc.mle_eval_base(nums, pt)) == list(cl[0])
    assert list(orc.mle_eval_ext(dens, pt)) == list(cl[1])
    bad = proof.copy()
    bad[40] ^= 1
    assert not ob.gkr_verify(orc, bad, log_n)[0]
