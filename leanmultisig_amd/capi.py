"""ctypes binding of include/leanmultisig.h (one-to-one), plus a thin Context helper that moves numpy arrays.

All field data are numpy uint32 arrays in Montgomery form.  Host EF arrays have a trailing axis of 5.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libleanmultisig_hip.so")

P = 0x7F000001
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class LmError(RuntimeError):
    pass


_SIGS = {
    "lm_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "lm_ctx_destroy": (None, [vp]),
    "lm_last_error": (C.c_char_p, []),
    "lm_sync": (C.c_int, [vp]),
    "lm_ctx_stream": (vp, [vp]),
    "lm_malloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "lm_free": (C.c_int, [vp, vp]),
    "lm_upload": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_download": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_memset_zero": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_ef_aos_to_soa": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_ef_soa_to_aos": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_poseidon16_permute": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_poseidon16_compress": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_commit": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(vp), vp]),
    "lm_tree_free": (None, [vp, vp]),
    "lm_tree_log_height": (C.c_uint32, [vp]),
    "lm_tree_leaf_words": (C.c_uint32, [vp]),
    "lm_tree_open": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp]),
    "lm_tree_download_matrix": (C.c_int, [vp, vp, vp]),
    "lm_tree_download_digests": (C.c_int, [vp, vp, vp]),
    "lm_mle_eval": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp]),
}

_lib = None


def load():
    """Load the HIP library.  Raises LmError if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LmError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; "
                      "leanmultisig_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(vp)


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a


class DeviceBuffer:
    """n_words of HBM owned by a Context."""

    def __init__(self, ctx, n_words):
        self.ctx = ctx
        self.n_words = int(n_words)
        p = vp()
        ctx._check(ctx.lib.lm_malloc(ctx.h, self.n_words, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = _u32(arr).reshape(-1)
        assert arr.size <= self.n_words
        self.ctx._check(self.ctx.lib.lm_upload(self.ctx.h, self.ptr, _ptr(arr), arr.size))
        return self

    def download(self, n_words=None, offset=0):
        n = self.n_words - offset if n_words is None else int(n_words)
        out = np.empty(n, dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_download(self.ctx.h, _ptr(out), self.ptr + 4 * offset, n))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.lm_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Tree:
    def __init__(self, ctx, handle, root):
        self.ctx, self.h, self.root = ctx, handle, root
        self.log_height = ctx.lib.lm_tree_log_height(handle)
        self.leaf_words = ctx.lib.lm_tree_leaf_words(handle)

    def open(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        n = idx.size
        leaves = np.empty((n, self.leaf_words), dtype=np.uint32)
        sib = np.empty((n, self.log_height, 8), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_tree_open(self.ctx.h, self.h, _ptr(idx), n, _ptr(leaves), _ptr(sib)))
        return leaves, sib

    def matrix(self):
        out = np.empty((1 << self.log_height, self.leaf_words), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_tree_download_matrix(self.ctx.h, self.h, _ptr(out)))
        return out

    def digests(self):
        out = np.empty(((2 << self.log_height) - 1, 8), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_tree_download_digests(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.lm_tree_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One lm_ctx (one GPU, one stream)."""

    def __init__(self, device=0):
        self.lib = load()
        h = vp()
        rc = self.lib.lm_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise LmError(f"lm_ctx_create({device}) -> {rc}: {self.lib.lm_last_error().decode()}")
        self.h = h.value

    def _check(self, rc):
        if rc != 0:
            raise LmError(f"leanmultisig error {rc}: {self.lib.lm_last_error().decode()}")

    def close(self):
        if self.h:
            self.lib.lm_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._check(self.lib.lm_sync(self.h))

    @property
    def stream(self):
        return self.lib.lm_ctx_stream(self.h)

    # ---- memory -------------------------------------------------------------------------------------
    def alloc(self, n_words):
        return DeviceBuffer(self, n_words)

    def to_device(self, arr):
        arr = _u32(arr).reshape(-1)
        return DeviceBuffer(self, max(arr.size, 1)).upload(arr)

    def ef_to_device_soa(self, ef_aos):
        """host (n,5) EF array -> device SoA buffer of 5n words"""
        a = _u32(ef_aos).reshape(-1, 5)
        return self.to_device(np.ascontiguousarray(a.T))

    # ---- ops ----------------------------------------------------------------------------------------
    def poseidon16(self, states, compress=False):
        st = _u32(states).reshape(-1, 16)
        buf = self.to_device(st)
        fn = self.lib.lm_poseidon16_compress if compress else self.lib.lm_poseidon16_permute
        self._check(fn(self.h, buf.ptr, st.shape[0]))
        return buf.download().reshape(-1, 16)

    def commit(self, d_evals, is_ext, n_vars, folding_factor, log_inv_rate, actual_len=None):
        if actual_len is None:
            actual_len = 1 << n_vars
        t = vp()
        root = np.empty(8, dtype=np.uint32)
        ptr = d_evals.ptr if isinstance(d_evals, DeviceBuffer) else int(d_evals)
        self._check(self.lib.lm_commit(self.h, ptr, int(bool(is_ext)), n_vars, folding_factor, log_inv_rate,
                                       int(actual_len), C.byref(t), _ptr(root)))
        return Tree(self, t.value, root)

    def mle_eval(self, d_evals, is_ext, n_vars, point, n_polys=1, stride_words=None):
        if stride_words is None:
            stride_words = (5 if is_ext else 1) << n_vars
        pt = _u32(point).reshape(-1)
        assert pt.size == n_vars * 5
        out = np.empty((n_polys, 5), dtype=np.uint32)
        ptr = d_evals.ptr if isinstance(d_evals, DeviceBuffer) else int(d_evals)
        self._check(self.lib.lm_mle_eval(self.h, ptr, int(bool(is_ext)), n_vars, n_polys, int(stride_words),
                                         _ptr(pt) if pt.size else None, _ptr(out)))
        return out
