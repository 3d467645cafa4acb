"""GPU parity: logup GKR (sum of fractions) through the C ABI vs the CPU oracle — transcripts must be word-identical —
with the instance family and checks of the reference's own test (quotient_gkr/mod.rs:222-301)."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,frac", [(6, 1.0), (7, 0.7), (11, 0.51), (11, 2 / 3), (13, 0.75), (13, 7 / 8), (15, 1.0)])
def test_gkr_matches_oracle(ctx, orc, log_n, frac):
    rng = np.random.default_rng(log_n * 7 + int(frac * 100))
    nums, dens = ob.gkr_instance(orc, rng, log_n, frac)
    ref_proof, rq, rpt, rcl = ob.gkr_prove(orc, nums, dens)
    pr = lm.Prover(ctx)
    q, pt, cl = pr.prove_gkr_quotient(ctx.to_device(nums), ctx.ef_to_device_soa(dens), log_n)
    proof = pr.proof()
    assert np.array_equal(q, rq) and np.array_equal(pt, rpt) and np.array_equal(cl, rcl)
    assert np.array_equal(proof, ref_proof)
    ok, vq, vpt, vcl, err = ob.gkr_verify(orc, proof, log_n)
    assert ok, err


@pytest.mark.parametrize("log_n,frac", [(6, 0.3), (7, 0.7), (9, 0.05), (11, 0.51), (11, 2 / 3), (13, 0.75), (13, 0.126), (15, 0.999), (12, 1.0)])
def test_gkr_active_prefix_matches_oracle(ctx, orc, log_n, frac):
    """lm_gkr_build_active: the (0, 1) tail is neither stored nor read (garbage is put there), its contribution enters in closed
    form — the reference's symbolic padding (sumcheck_utils.rs:136,225,331).  Same transcript as the fully materialised run."""
    rng = np.random.default_rng(1000 + log_n * 7 + int(frac * 100))
    nums, dens = ob.gkr_instance(orc, rng, log_n, frac)
    active = max(1, int((1 << log_n) * frac))
    ref_proof, rq, rpt, rcl = ob.gkr_prove(orc, nums, dens)
    tail = (active + 7) & ~7                         # what must exist in memory: the prefix rounded up to 8 entries
    junk_n, junk_d = nums.copy(), dens.copy()
    junk_n[tail:] = ob.rand_field(rng, junk_n[tail:].shape)
    junk_d[tail:] = ob.rand_field(rng, junk_d[tail:].shape)
    pr = lm.Prover(ctx)
    q, pt, cl = pr.prove_gkr_quotient(ctx.to_device(junk_n), ctx.ef_to_device_soa(junk_d), log_n, active_len=active)
    assert np.array_equal(q, rq) and np.array_equal(pt, rpt) and np.array_equal(cl, rcl)
    assert np.array_equal(pr.proof(), ref_proof)


def test_gkr_large_verifies_and_claims_match_mle(ctx, orc):
    """2^20 entries: too slow for the oracle prover; checked by the oracle verifier + device MLE evaluations of the
    inputs at the returned point (the two checks of the reference test, mod.rs:276-279)."""
    log_n = 20
    rng = np.random.default_rng(1)
    nums, dens = ob.gkr_instance(orc, rng, log_n, 0.8)
    d_n, d_d = ctx.to_device(nums), ctx.ef_to_device_soa(dens)
    pr = lm.Prover(ctx)
    q, pt, cl = pr.prove_gkr_quotient(d_n, d_d, log_n)
    ok, vq, vpt, vcl, err = ob.gkr_verify(orc, pr.proof(), log_n)
    assert ok, err
    assert np.array_equal(vq, q) and np.array_equal(vpt, pt) and np.array_equal(vcl, cl)
    assert list(ctx.mle_eval(d_n, False, log_n, pt)[0]) == list(cl[0])
    assert list(ctx.mle_eval(d_d, True, log_n, pt)[0]) == list(cl[1])


@pytest.mark.parametrize("log_n,rounds_before_drop", [(14, 1), (14, 3), (14, 7), (9, 1), (12, 5)])
def test_gkr_abandoned_layer_dismisses_the_resident_tail(ctx, orc, log_n, rounds_before_drop):
    """The tail of a layer's sumcheck stays resident on the device between challenge pairs (k_gkr_tail: up to 16 workgroups
    polling a mailbox).  A caller that drops the layer half way — error path, early free — must not leave them polling: lm_gkr_free
    dismisses them (well inside their own 3 s timeout), the hand-over ticket is re-armed, and the context proves the next
    instance as if nothing had happened."""
    import ctypes as C
    import time
    rng = np.random.default_rng(4242 + log_n + rounds_before_drop)
    nums, dens = ob.gkr_instance(orc, rng, log_n, 0.9)
    d_n, d_d = ctx.to_device(nums), ctx.ef_to_device_soa(dens)
    lib = ctx.lib
    g = C.c_void_p()
    assert lib.lm_gkr_build(ctx.h, d_n.ptr, d_d.ptr, log_n, C.byref(g)) == 0
    K = log_n - 1  # the input layer: 2^(K+1) entries, K rounds
    point = ob.rand_field(rng, (K, 5)).astype(np.uint32)
    alpha = ob.rand_field(rng, (5,)).astype(np.uint32)
    assert lib.lm_gkr_layer_begin(ctx.h, g, K, point.ctypes.data_as(C.c_void_p), alpha.ctypes.data_as(C.c_void_p)) == 0
    out = np.zeros(10, dtype=np.uint32)
    r = None
    for t in range(rounds_before_drop):
        assert lib.lm_gkr_round(ctx.h, g, None if r is None else r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        r = ob.rand_field(rng, (5,)).astype(np.uint32)
    t0 = time.time()
    lib.lm_gkr_free(ctx.h, g)
    assert time.time() - t0 < 1.0
    # the same context proves a fresh instance correctly afterwards
    nums2, dens2 = ob.gkr_instance(orc, rng, 13, 0.75)
    ref_proof, rq, rpt, rcl = ob.gkr_prove(orc, nums2, dens2)
    pr = lm.Prover(ctx)
    q, pt, cl = pr.prove_gkr_quotient(ctx.to_device(nums2), ctx.ef_to_device_soa(dens2), 13)
    assert np.array_equal(q, rq) and np.array_equal(pt, rpt) and np.array_equal(cl, rcl)
    assert np.array_equal(pr.proof(), ref_proof)


def test_gkr_launch_per_round_pair_schedule_matches_oracle():
    """LM_GKR_NO_TAIL=1 — one k_gkr_step launch per round pair down to the last entries, which is also what a layer falls back to
    when too many resident tails are alive in the process — must give the same transcripts (the switch is read once per process,
    so the parity tests above are re-run in a child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LM_GKR_NO_TAIL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gkr_gpu.py", "-k",
                        "test_gkr_matches_oracle or test_gkr_active_prefix_matches_oracle"], env=env, cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "16 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("env_name,env_value", [("LM_GKR_NO_AHEAD", "1"), ("LM_GKR_TAIL_W", "64"), ("LM_GKR_TAIL_W", "2")])
def test_gkr_schedule_variants_match_oracle(env_name, env_value):
    """The launches of a layer are enqueued AHEAD of their challenges by default (k_gkr_step / k_gkr_tail wait for message n of the
    launch-ahead line, lm_mail_*): LM_GKR_NO_AHEAD=1 restores one launch per exchange.  LM_GKR_TAIL_W sets the workgroups of a resident
    tail (default 16; 64: layers enter it at 2^14 entries and the hand-over gathers 64 slices; 2: almost every round pair is a launch).
    Same transcripts either way (the switches are read once per process: the parity tests are re-run in a child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{env_name: env_value})
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gkr_gpu.py", "-k",
                        "test_gkr_matches_oracle or test_gkr_active_prefix_matches_oracle or test_gkr_abandoned_layer or test_gkr_large"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


_FAIL_SOFT_CHILD = """
import numpy as np
import leanmultisig_amd as lm
from tests import oracle_binding as ob
orc = ob.load()
ctx = lm.Context(0)
rng = np.random.default_rng(5)
nums, dens = ob.gkr_instance(orc, rng, 15, 0.8)
ref_proof, rq, rpt, rcl = ob.gkr_prove(orc, nums, dens)
d_n, d_d = ctx.to_device(nums), ctx.ef_to_device_soa(dens)
for rep in range(2):
    pr = lm.Prover(ctx)
    q, pt, cl = pr.prove_gkr_quotient(d_n, d_d, 15, active_len=int(0.8 * (1 << 15)))
    assert np.array_equal(q, rq) and np.array_equal(pt, rpt) and np.array_equal(cl, rcl)
    assert np.array_equal(pr.proof(), ref_proof), "proof differs after the fallback"
print("FALLBACKS", ctx.soft_fallbacks())
"""


@pytest.mark.parametrize("fault", ["tail:1", "tail:6", "ahead:1", "ahead:2"])
def test_gkr_fails_soft_when_a_resident_kernel_never_gets_its_message(fault):
    """A resident tail / a launch enqueued ahead that is starved of its wave slots on a shared device looks to the host like a kernel whose
    message never arrives: it gives up after its 3 s, nothing is published.  LM_GKR_FAULT drops the n-th message of that kind; lm_gkr_round
    must dismiss the resident kernels, re-run the layer from its storage with one launch per exchange and deliver the SAME transcript —
    an internal scheduling event (lm_soft_fallbacks), not a prover error."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LM_GKR_FAULT=fault, LM_GKR_AHEAD_ASSUME_ALONE="1")
    r = subprocess.run([sys.executable, "-c", _FAIL_SOFT_CHILD], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "FALLBACKS 1" in r.stdout, r.stdout[-500:]   # the first proof falls back; the context then keeps off resident kernels for 2 s, so the second has none to lose


def test_gkr_without_fault_never_falls_back(ctx, orc):
    rng = np.random.default_rng(6)
    nums, dens = ob.gkr_instance(orc, rng, 15, 1.0)
    before = ctx.soft_fallbacks()
    pr = lm.Prover(ctx)
    pr.prove_gkr_quotient(ctx.to_device(nums), ctx.ef_to_device_soa(dens), 15)
    assert ctx.soft_fallbacks() == before
