"""ctypes binding of the CPU oracle (oracle/liblm_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liblm_oracle.so")
P = 0x7F000001
vp = C.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return LIB


def _p(a):
    return a.ctypes.data_as(vp)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        for n in ("orc_to_monty", "orc_from_monty", "orc_add", "orc_sub", "orc_mul", "orc_inv", "orc_two_adic_generator"):
            getattr(lib, n).restype = C.c_uint32
            getattr(lib, n).argtypes = [C.c_uint32] * (1 if n in ("orc_to_monty", "orc_from_monty", "orc_inv", "orc_two_adic_generator") else 2)
        lib.orc_merkle_verify.restype = C.c_int

    # field helpers (vectorised with numpy where trivial)
    def to_monty(self, x):
        x = np.asarray(x, dtype=np.uint64)
        return ((x << np.uint64(32)) % np.uint64(P)).astype(np.uint32)

    def from_monty(self, x):
        x = np.asarray(x, dtype=np.uint32)
        return np.array([self.lib.orc_from_monty(int(v)) for v in x.reshape(-1)], dtype=np.uint32).reshape(x.shape)

    def mul(self, a, b):
        return self.lib.orc_mul(int(a), int(b))

    def ef_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        o = np.empty(5, dtype=np.uint32)
        self.lib.orc_ef_mul(_p(a), _p(b), _p(o))
        return o

    def ef_inv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty(5, dtype=np.uint32)
        self.lib.orc_ef_inv(_p(a), _p(o))
        return o

    def poseidon16_permute(self, states):
        s = np.array(states, dtype=np.uint32).reshape(-1, 16).copy()
        self.lib.orc_poseidon16_permute(_p(s), C.c_uint64(s.shape[0]))
        return s

    def poseidon16_compress(self, states):
        s = np.array(states, dtype=np.uint32).reshape(-1, 16).copy()
        self.lib.orc_poseidon16_compress(_p(s), C.c_uint64(s.shape[0]))
        return s

    def hash_slice(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint32)
        o = np.empty(8, dtype=np.uint32)
        self.lib.orc_hash_slice(_p(d), C.c_uint64(d.size), _p(o))
        return o

    def lde_base(self, evals, fold, log_inv_rate, dft_n_cols=None):
        e = np.ascontiguousarray(evals, dtype=np.uint32)
        n_cols = (1 << fold) if dft_n_cols is None else dft_n_cols
        h = (e.size << log_inv_rate) >> fold
        out = np.empty((h, n_cols), dtype=np.uint32)
        self.lib.orc_lde_base(_p(e), C.c_uint64(e.size), C.c_uint32(fold), C.c_uint32(log_inv_rate), C.c_uint64(n_cols), _p(out))
        return out

    def lde_ext(self, evals, fold, log_inv_rate, dft_n_cols=None):
        e = np.ascontiguousarray(evals, dtype=np.uint32).reshape(-1, 5)
        n_cols = (1 << fold) if dft_n_cols is None else dft_n_cols
        h = (e.shape[0] << log_inv_rate) >> fold
        out = np.empty((h, n_cols * 5), dtype=np.uint32)
        self.lib.orc_lde_ext(_p(e), C.c_uint64(e.shape[0]), C.c_uint32(fold), C.c_uint32(log_inv_rate), C.c_uint64(n_cols), _p(out))
        return out

    def merkle_build(self, rows, full_width):
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        h, w = r.shape
        out = np.empty((2 * h - 1, 8), dtype=np.uint32)
        self.lib.orc_merkle_build(_p(r), C.c_uint64(h), C.c_uint64(w), C.c_uint64(full_width), _p(out))
        return out

    def merkle_verify(self, root, log_height, index, leaf, siblings):
        root = np.ascontiguousarray(root, dtype=np.uint32)
        leaf = np.ascontiguousarray(leaf, dtype=np.uint32)
        sib = np.ascontiguousarray(siblings, dtype=np.uint32)
        return bool(self.lib.orc_merkle_verify(_p(root), C.c_uint64(log_height), C.c_uint64(index), _p(leaf),
                                               C.c_uint64(leaf.size), _p(sib)))

    def mle_eval_base(self, v, point):
        v = np.ascontiguousarray(v, dtype=np.uint32)
        pt = np.ascontiguousarray(point, dtype=np.uint32).reshape(-1, 5)
        assert v.size == 1 << pt.shape[0]
        o = np.empty(5, dtype=np.uint32)
        self.lib.orc_mle_eval_base(_p(v), C.c_uint32(pt.shape[0]), _p(pt), _p(o))
        return o

    def mle_eval_ext(self, v, point):
        v = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1, 5)
        pt = np.ascontiguousarray(point, dtype=np.uint32).reshape(-1, 5)
        assert v.shape[0] == 1 << pt.shape[0]
        o = np.empty(5, dtype=np.uint32)
        self.lib.orc_mle_eval_ext(_p(v), C.c_uint32(pt.shape[0]), _p(pt), _p(o))
        return o

    def expand_from_univariate(self, a, n):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty((n, 5), dtype=np.uint32)
        self.lib.orc_expand_from_univariate(_p(a), C.c_uint32(n), _p(o))
        return o

    def eq_table(self, point, scalar=None):
        pt = np.ascontiguousarray(point, dtype=np.uint32).reshape(-1, 5)
        if scalar is None:
            scalar = np.array([0x01FFFFFE, 0, 0, 0, 0], dtype=np.uint32)
        s = np.ascontiguousarray(scalar, dtype=np.uint32)
        o = np.empty((1 << pt.shape[0], 5), dtype=np.uint32)
        self.lib.orc_eq_table(_p(pt), C.c_uint32(pt.shape[0]), _p(s), _p(o))
        return o


_inst = None


def load():
    global _inst
    if _inst is None:
        build()
        _inst = Oracle(C.CDLL(LIB))
    return _inst


def rand_field(rng, shape):
    """i.i.d. uniform F_p elements as Montgomery-form words (any word < p is a valid Montgomery value)."""
    return rng.integers(0, P, size=shape, dtype=np.uint32)
