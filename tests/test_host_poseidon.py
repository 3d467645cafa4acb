"""CPU tests of the host-side Poseidon1-16 (csrc/host/lm_poseidon_x86.cpp): the permutation every transcript operation runs.
The AVX-512 backend (when this CPU has it) must be the same function as the scalar code, which is pinned by the reference's
KAT (poseidon1_koalabear_16.rs:1083-1091) here and against the oracle in test_host_mirrors.  On a CPU without AVX-512 IFMA
both calls run the scalar code and the comparison is trivially true; the KAT still checks it."""
import numpy as np

import leanmultisig_amd as lm
from tests.oracle_binding import P
from tests.test_oracle_pins import KAT_OUT


def test_backend_is_reported():
    assert lm.host_poseidon_backend() in ("avx512-ifma", "scalar")


def test_kat_both_backends(orc):
    st = orc.to_monty(np.arange(16))
    for scalar in (False, True):
        out = lm.host_poseidon16_permute(st, scalar=scalar)
        assert list(orc.from_monty(out)) == KAT_OUT


def test_backend_equals_scalar_and_oracle(orc):
    rng = np.random.default_rng(11)
    states = [np.zeros(16, dtype=np.uint32), np.full(16, P - 1, dtype=np.uint32), np.full(16, 1, dtype=np.uint32)]
    e = np.zeros(16, dtype=np.uint32)
    for i in range(16):  # single extreme words: carries in every lane position of the vector code
        v = e.copy()
        v[i] = P - 1
        states.append(v)
    states += [rng.integers(0, P, size=16, dtype=np.uint32) for _ in range(3000)]
    # a sponge-like chain: output fed back (values far from uniform inputs are reached too)
    s = states[-1]
    for _ in range(2000):
        a, b = lm.host_poseidon16_permute(s), lm.host_poseidon16_permute(s, scalar=True)
        assert np.array_equal(a, b)
        s = a
    got = np.stack([lm.host_poseidon16_permute(x) for x in states])
    ref = np.stack([lm.host_poseidon16_permute(x, scalar=True) for x in states])
    assert np.array_equal(got, ref)
    assert np.array_equal(got, orc.poseidon16_permute(np.stack(states)))
