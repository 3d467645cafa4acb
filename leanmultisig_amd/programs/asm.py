"""Function-body helper over leanmultisig_amd.vm.Program: frame cells, address operands, extension-field and Poseidon macros.

The reference writes its recursion program in a zkDSL (crates/rec_aggregation/zkdsl_implem/*.py) and compiles it
(crates/lean_compiler, out of scope here); programs in this package are written at the ISA level.  This module holds what every
such program repeats: a bump allocator of fp-relative cells (memory is write-once: every temporary is a fresh cell), the three
kinds of address a precompile operand can carry, and the lowerings of the DSL's helpers that are single instructions
(utils.py:250-361: mul_extension, add_extension, sub_extension, copy_5, dot_product_*, poly_eq_*).

Locations (an ADDRESS, not a value):   fp(off) = fp + off   ·   absolute(a)   ·   at(cell, k) = m[fp + cell] + k
ISA rule for precompiles (instruction_encoder.rs:60-113): operands a, b are BOTH fp-relative or each a constant / a cell holding
the address; the result operand may be any of the three.  `Fn.ab` inserts the 1-cycle `cell = fp + off` where a mixed pair needs it.
"""
from ..vm import FP, K, M, Label

DIM, DIGEST_LEN = 5, 8


class Loc:
    __slots__ = ("kind", "a", "k")

    def __init__(self, kind, a, k=0):
        self.kind, self.a, self.k = kind, a, k

    def __add__(self, n):
        n = int(n)
        if self.kind == "at":
            return Loc("at", self.a, self.k + n)
        return Loc(self.kind, self.a + n)

    def __repr__(self):
        return f"{self.kind}({self.a}{', ' + str(self.k) if self.kind == 'at' else ''})"


def fp(off):
    return Loc("fp", int(off))


def absolute(a):
    return Loc("abs", int(a))


def at(cell, k=0):
    return Loc("at", int(cell), int(k))


class Fn:
    """the body of one function: `p` the Program, cells allocated from `start` upwards"""

    def __init__(self, p, start=0, zero_vec_ptr=None, one_ef_ptr=None, ones_ptr=None):
        self.p, self.top = p, start
        self.ZERO, self.ONE, self.ONES = zero_vec_ptr, one_ef_ptr, ones_ptr
        self._ptr = {}        # (kind, a, k) -> cell holding that address

    # ---- cells ----------------------------------------------------------------------------------------------------------------------
    def alloc(self, n=1):
        o = self.top
        self.top += n
        return o

    def fresh(self, stem):
        """a label name no other function body of this program uses"""
        n = self.p.__dict__.setdefault("_fresh_labels", 0) + 1
        self.p._fresh_labels = n
        return f"{stem}@{n}"

    def const(self, value):
        c = self.alloc()
        self.p.add(K(0), K(value), M(c))
        return c

    def scope(self):
        """pointer cells created inside a jump-table entry are not defined on the other paths: snapshot / restore the cache"""
        fn = self

        class _S:
            def __enter__(self_s):
                self_s.saved = dict(fn._ptr)

            def __exit__(self_s, *a):
                fn._ptr = self_s.saved
        return _S()

    def forget_pointers(self):
        self._ptr = {}

    # ---- addresses ------------------------------------------------------------------------------------------------------------------
    def ptr(self, loc):
        """a cell holding the address `loc` (cached within straight-line code)"""
        if loc.kind == "at" and loc.k == 0:
            return loc.a
        key = (loc.kind, loc.a, loc.k)
        c = self._ptr.get(key)
        if c is None:
            c = self.alloc()
            if loc.kind == "fp":
                self.p.add(K(0), FP(loc.a), M(c))
            elif loc.kind == "abs":
                self.p.add(K(0), K(loc.a), M(c))
            else:
                self.p.add(M(loc.a), K(loc.k), M(c))
            self._ptr[key] = c
        return c

    def _one(self, loc):
        return K(loc.a) if loc.kind == "abs" else M(self.ptr(loc))

    def ab(self, a, b):
        if a.kind == "fp" and b.kind == "fp":
            return FP(a.a), FP(b.a)
        return self._one(a), self._one(b)

    def res(self, loc):
        if loc.kind == "fp":
            return FP(loc.a)
        return self._one(loc)

    # ---- words ----------------------------------------------------------------------------------------------------------------------
    def load(self, loc):
        """-> a cell holding the word at `loc`"""
        if loc.kind == "fp":
            return loc.a
        c = self.alloc()
        if loc.kind == "at":
            self.p.deref(loc.a, loc.k, M(c))
        else:
            self.p.deref(self.ptr(absolute(0)), loc.a, M(c))
        return c

    def store(self, loc, operand):
        """m[loc] = operand (an assertion when the cell is already defined)"""
        if loc.kind == "fp":
            self.p.add(K(0), operand, M(loc.a))
        elif loc.kind == "at":
            self.p.deref(loc.a, loc.k, operand)
        else:
            self.p.deref(self.ptr(absolute(0)), loc.a, operand)

    def assert_zero(self, loc):
        self.store(loc, K(0))

    # ---- extension field (one ExtensionOp instruction each) ----------------------------------------------------------------------------
    def ext(self, op, a, b, dst, size=1, be=False):
        oa, ob = self.ab(a, b)
        self.p.extension_op(op, oa, ob, self.res(dst), size=size, is_be=be)

    def new_ef(self, n=1):
        return fp(self.alloc(DIM * n))

    def mul(self, a, b, dst=None):
        dst = dst or self.new_ef()
        self.ext("mul", a, b, dst)
        return dst

    def add(self, a, b, dst=None):
        dst = dst or self.new_ef()
        self.ext("add", a, b, dst)
        return dst

    def sub(self, a, b, dst=None):
        """dst = a - b: add_ee(b, dst, a) solves its unknown operand (extension_op/exec.rs:29-94; utils.py:319-323)"""
        dst = dst or self.new_ef()
        self.ext("add", b, dst, a)
        return dst

    def copy5(self, src, dst):
        """dot_product_ee(src, ONE_EF_PTR, dst) (utils.py:359-361)"""
        self.ext("mul", src, absolute(self.ONE), dst)

    def copy8(self, src, dst):
        """copy_8 (utils.py:416-420): two overlapping copies of five"""
        self.copy5(src, dst)
        self.copy5(src + 3, dst + 3)

    def dot(self, a, b, dst, n, be=False):
        self.ext("dot_product", a, b, dst, size=n, be=be)
        return dst

    def poly_eq(self, a, b, dst, n, be=False):
        self.ext("poly_eq", a, b, dst, size=n, be=be)
        return dst

    def set_one(self, loc):
        self.copy5(absolute(self.ONE), loc)

    def powers(self, alpha, n):
        """powers_const (utils.py:49-61): [1, alpha, ..., alpha^(n-1)], n >= 1"""
        res = self.new_ef(n)
        self.set_one(res)
        if n > 1:
            self.copy5(alpha, res + DIM)
            for i in range(1, n - 1):
                self.mul(res + i * DIM, res + DIM, res + (i + 1) * DIM)
        return res

    def eq_mle(self, point, n):
        """compute_eq_mle_extension (utils.py:86-102): the 2^n values of eq(point, .), first coordinate = most significant index bit"""
        if n == 0:
            return absolute(self.ONE)
        res = self.new_ef((2 << n) - 1)
        self.set_one(res)
        for s in range(n):
            pt = self.new_ef()
            self.copy5(point + (n - 1 - s) * DIM, pt)
            for i in range(1 << s):
                hi = res + ((2 << s) - 1 + (1 << s) + i) * DIM
                self.mul(pt, res + ((1 << s) - 1 + i) * DIM, hi)
                self.sub(res + ((1 << s) - 1 + i) * DIM, hi, res + ((2 << s) - 1 + i) * DIM)
        return res + ((1 << n) - 1) * DIM

    # ---- Poseidon16 ---------------------------------------------------------------------------------------------------------------------
    def compress(self, a, b, dst):
        oa, ob = self.ab(a, b)
        self.p.poseidon16(oa, ob, self.res(dst))

    def permute(self, a, b, dst):
        oa, ob = self.ab(a, b)
        self.p.poseidon16(oa, ob, self.res(dst), permute=True)

    # ---- control flow ---------------------------------------------------------------------------------------------------------------------
    def range_check(self, cell, bound):
        """m[fp + cell] <= bound (3 cycles, b_compile_intermediate.rs:673-752)"""
        self.p.range_check(cell, K(bound) if not hasattr(bound, "mode") else bound, self.alloc(3))

    def dispatch(self, index_cell, block, table_label):
        """jump to table_label + m[index] * block; the table's entries jump back to the returned label (match_range)"""
        off, dest = self.alloc(), self.alloc()
        self.p.mul(M(index_cell), K(block), M(off))
        self.p.add(M(off), K(Label(table_label)), M(dest))
        self.p.jump(K(1), M(dest), FP(0))
        after = table_label + "@after"
        self.p.label(after)
        return after

    def call_loop(self, label, size_label, args):
        """first call of a (parallel) loop function: frame [return pc, saved fp, 0, args...] (main.py:161-167 lowering of a loop)"""
        frame = self.alloc()
        self.p.hint_request_memory(frame, K(Label(size_label)))
        back = self.fresh("after_" + label)
        self.p.deref(frame, 0, K(Label(back)))
        self.p.deref(frame, 1, FP(0))
        self.p.deref(frame, 2, K(0))
        for k, a in enumerate(args):
            self.p.deref(frame, 3 + k, a)
        self.p.jump(K(1), K(Label(label)), M(frame))
        self.p.label(back)


def loop_prologue(p, g, label, n_args, parallel=True):
    """the head of a loop function whose frame is [return pc, saved fp, i, end, args...]: returns when i == end.  With `parallel` the
    runner hands iterations 1.. to segments (Hint::ParallelBatchStart, runner.rs:369-482)."""
    I, END = 2, 3
    if parallel:
        p.hint_parallel_batch_start(n_args, M(END))
    p.label(label)
    d, dinv, nz, omnz = g.alloc(), g.alloc(), g.alloc(), g.alloc()
    p.add(M(d), M(END), M(I))
    p.hint_inverse(M(d), dinv)
    p.mul(M(d), M(dinv), M(nz))
    p.add(M(omnz), M(nz), K(1))
    p.mul(M(omnz), M(d), K(0))
    body = label + "@body"
    p.jump(M(nz), K(Label(body)), FP(0))
    p.jump(K(1), M(0), M(1))
    p.label(body)


def loop_epilogue(p, g, label, size_label, n_args):
    """allocate the next iteration's frame, copy the arguments with i + 1, and enter it"""
    nxt, ip1 = g.alloc(), g.alloc()
    p.hint_request_memory(nxt, K(Label(size_label)))
    p.deref(nxt, 0, M(0))
    p.deref(nxt, 1, M(1))
    p.add(M(2), K(1), M(ip1))
    p.deref(nxt, 2, M(ip1))
    for a in range(3, 2 + n_args):
        p.deref(nxt, a, M(a))
    p.jump(K(1), K(Label(label)), M(nxt))
