#!/usr/bin/env python3
"""Differential fuzzing of the leanVM runners: seeded random programs over the assembler (leanmultisig_amd/vm.py) — a head in main, ONE
parallel loop whose body is a random sequence of instructions / hints / precompile calls, optionally a second parallel loop and a
sequential tail — executed by
    the host runner with its Poseidon calls executed at once          (lmh_execute_bytecode, LM_VM_LAZY=0)
    the host runner with recorded / deferred precompile calls          (LM_VM_LAZY=1: csrc/host/lm_vm.cpp MemBuf)
    the device runner, batches as wavefronts                           (lmh_execute_bytecode_device; only with a context)
    the oracle's sequential restatement of runner.rs                   (oracle/vm_oracle.hpp)
and compared on RESULTS (pc / fp per cycle, memory image, defined mask, instruction counts, precompile records) and on ERRORS (the
library's three paths: the same text; against the oracle: the same RunnerError kind).  A program is valid by construction (write-once
cells, defined reads) unless its seed asks for a fault: an assertion that fails in one iteration only (a zero in the hinted data), two
iterations writing one cell, a read of a cell nothing defines, an ExtensionOp check that does not hold.

What a body draws from: ADD / MUL in every operand mode incl. the solved-unknown forms; DEREF loads (hinted data, own frame through a
pointer), stores (own frame, deferred writes into main's arrays at a permuted slot, into the NEXT iteration's spare argument cells) and
the range-check pattern (DerefHint: resolved after the run); conditional jumps and 4-entry jump tables; Poseidon16 compress / half
output / hard-coded left / permute with fp-relative, pointer and constant operands and results inside or outside the frame;
ExtensionOp add / dot_product / poly_eq, base-by-extension or not, lengths 1-6, solved-unknown forms; every hint (inverse,
decompose_bits_xmss / _merkle_whir / decompose_bits, less_than, log2_ceil, witness inline / indirect, debug_assert); reads of a digest
main's head has not computed yet (the deferred-call path) early or late in the body.

    python tools/vm_fuzz.py --seeds 10000            # CPU: host eager == host lazy == oracle
    python tools/vm_fuzz.py --seeds 1000 --device    # adds the device runner (33-48 iterations per loop)
"""
import argparse
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leanmultisig_amd import vm  # noqa: E402
from leanmultisig_amd.vm import FP, K, M, Label, Program, Witness, to_monty  # noqa: E402

P = 0x7F000001
OUTW = 40            # words of main's OUT array one iteration may define
MAIN_FRAME = 160
HEAD_BLOCK, HEAD_DIG = 64, 80     # main cells: a hinted 16-word block, the digests of the head's chain
TWEAKS = 8                        # absolute address of 4-word "tweaks" for the hard-coded-left variant: the public input words 8.. do not
                                  # exist, so main's first cells are used (see gen)


def mont(x):
    return to_monty(np.asarray(x)).astype(np.uint32)


class Body:
    """one loop body: cells of the frame [ret, fp, i, end, OUT, DATA, PERM, DIG | d, inv, nz, omnz | locals]"""

    def __init__(self, p, rng, n_args, tag, n_iter, faults):
        self.p, self.rng, self.tag, self.n_iter = p, rng, tag, n_iter
        self.top = 2 + n_args + 4
        self.words = [2]          # cells holding a defined word (the iteration counter to start with)
        self.small = [2]          # ... known to be < n_iter
        self.efs = []             # frame offsets of 5 defined words
        self.blocks = []          # frame offsets of 8 defined words
        self.out_slot = 0 if tag == "_a" else OUTW // 2   # next free word of this iteration's OUT row (each loop owns half a row)
        self.out_end = OUTW // 2 if tag == "_a" else OUTW
        self.hints = {}           # per-iteration inline hint streams: name -> words per entry
        self.faults = faults
        self.tables = []
        self.nlabel = 0

    def alloc(self, n=1):
        o = self.top
        self.top += n
        return o

    def label(self, stem):
        self.nlabel += 1
        return f"{stem}_{self.tag}_{self.nlabel}"

    def word(self):
        return int(self.rng.choice(self.words))

    def out_ptr(self, n):
        """a cell holding the address of `n` fresh words of this iteration's OUT row (perm[i] * OUTW + slot)"""
        p = self.p
        if self.out_slot + n > self.out_end:
            return None
        if not hasattr(self, "_row"):
            pp, pi, o, row = self.alloc(), self.alloc(), self.alloc(), self.alloc()
            p.add(M(6), M(2), M(pp))
            p.deref(pp, 0, M(pi))                      # perm[i]
            p.mul(M(pi), K(OUTW), M(o))
            p.add(M(4), M(o), M(row))
            self._row = row
            self.small.append(pi)
            self.words.append(pi)
        c = self.alloc()
        p.add(M(self._row), K(self.out_slot), M(c))
        self.out_slot += n
        return c

    def frame_ptr(self, off):
        c = self.alloc()
        self.p.add(K(0), FP(off), M(c))
        return c


def op_arith(b):
    p, rng = b.p, b.rng
    a = K(int(rng.integers(0, P))) if rng.random() < 0.3 else M(b.word())
    r = rng.random()
    c = K(int(rng.integers(0, P))) if r < 0.3 else (FP(int(rng.integers(0, 200))) if r < 0.45 else M(b.word()))
    new = b.alloc()
    (p.mul if rng.random() < 0.5 else p.add)(a, c, M(new))
    b.words.append(new)


def op_solve(b):
    """the unknown operand of an ADD / MUL is solved for (instruction.rs:146-200)"""
    p, rng = b.p, b.rng
    x, y, new = b.word(), b.word(), b.alloc()
    form = int(rng.integers(0, 4))
    if form == 0:
        p.add(M(new), M(x), M(y))          # new = y - x
    elif form == 1:
        p.add(M(x), M(new), M(y))
    else:
        nz = b.alloc()                     # a divisor that is never zero: x^2 + 1 has no root (p = 3 mod 4)
        sq = b.alloc()
        p.mul(M(x), M(x), M(sq))
        p.add(M(sq), K(1), M(nz))
        b.words.append(nz)
        if form == 2:
            p.mul(M(new), M(nz), M(y))
        else:
            p.mul(M(nz), M(new), M(y))
    b.words.append(new)


def op_assert(b):
    p = b.p
    x, y = b.word(), b.alloc()
    p.add(M(x), K(0), M(y))
    p.add(M(y), K(0), M(x))                # all three known: a check
    b.words.append(y)


def op_load_data(b):
    p, rng = b.p, b.rng
    new = b.alloc()
    p.deref(5, int(rng.integers(0, b.data_words)), M(new))
    b.words.append(new)


def op_store_out(b):
    c = b.out_ptr(1)
    if c is None:
        return
    b.p.deref(c, 0, M(b.word()))


def op_frame_pointer_store_load(b):
    p = b.p
    cell, got = b.alloc(), b.alloc()
    ptr = b.frame_ptr(cell)
    p.deref(ptr, 0, M(b.word()))           # a store through a pointer into the own frame
    p.deref(ptr, 0, M(got))                # and the load back
    b.words += [cell, got]


def op_range_check(b):
    x = int(b.rng.choice(b.small))
    b.p.range_check(x, K(b.n_iter + 7), b.alloc(3))


def op_branch(b):
    """if less_than(x, y): z = 1 + x else z = 2 * y — a conditional jump on a hinted, checked boolean"""
    p = b.p
    x, y, bit, z = b.word(), b.word(), b.alloc(), b.alloc()
    p.hint_less_than(M(x), M(y), M(bit))
    p.mul(M(bit), M(bit), M(bit))                  # boolean
    then, end = b.label("then"), b.label("end")
    p.jump(M(bit), K(Label(then)), FP(0))
    p.mul(M(y), K(2), M(z))
    p.jump(K(1), K(Label(end)), FP(0))
    p.label(then)
    p.add(M(x), K(1), M(z))
    p.label(end)
    b.words.append(z)


def op_table(b):
    """match_range over the two low bits of a small value"""
    p, rng = b.p, b.rng
    x = int(rng.choice(b.small))
    nb = 8
    bits = b.alloc(nb)
    p.hint_decompose_bits(M(x), FP(bits), K(nb))   # big-endian
    v, off, dest, y = b.alloc(), b.alloc(), b.alloc(), b.alloc()
    t = b.alloc()
    p.mul(M(bits + nb - 2), K(2), M(t))
    p.add(M(t), M(bits + nb - 1), M(v))
    p.mul(M(v), K(2), M(off))
    tab, after = b.label("table"), b.label("after")
    p.add(M(off), K(Label(tab)), M(dest))
    p.jump(K(1), M(dest), FP(0))
    p.label(after)
    b.tables.append((tab, after, y, [int(rng.integers(0, P)) for _ in range(4)]))
    b.words += [v, y]
    b.small.append(v)


def _block(b):
    """an operand that addresses 8 defined words: (operand kind, value)"""
    r = b.rng.random()
    if r < 0.55 or not b.blocks:
        return ("fp", int(b.rng.choice(b.blocks))) if b.blocks else ("data", int(b.rng.integers(0, b.data_words - 8)))
    if r < 0.85:
        return ("data", int(b.rng.integers(0, b.data_words - 8)))
    return ("abs", 0)                              # the public input


def _operands(b, x, y):
    """precompile operands a, b: both fp-relative or neither (instruction_encoder.rs)"""
    p = b.p

    def one(kv):
        kind, v = kv
        if kind == "abs":
            return K(v)
        c = b.alloc()
        if kind == "fp":
            p.add(K(0), FP(v), M(c))
        else:
            p.add(M(5), K(v), M(c))
        return M(c)
    if x[0] == "fp" and y[0] == "fp":
        return FP(x[1]), FP(y[1])
    return one(x), one(y)


def op_poseidon(b):
    p, rng = b.p, b.rng
    variant = rng.choice(["compress", "half", "left", "permute"], p=[0.4, 0.2, 0.2, 0.2])
    n_out = {"compress": 8, "half": 4, "left": 8, "permute": 16}[variant]
    a, c = _operands(b, _block(b), _block(b))
    outside = rng.random() < 0.25
    dst = None
    if outside:
        oc = b.out_ptr(n_out)
        if oc is not None:
            dst = M(oc)
    if dst is None:
        o = b.alloc(n_out)
        dst = FP(o)
        if n_out >= 8:
            b.blocks.append(o)
            if n_out == 16:
                b.blocks.append(o + 8)
        if n_out >= 5:
            b.efs.append(o)
        b.words.append(o)
    if variant == "compress":
        p.poseidon16(a, c, dst)
    elif variant == "half":
        p.poseidon16(a, c, dst, half=True)
    elif variant == "left":
        p.poseidon16(a, c, dst, half=bool(rng.random() < 0.5) and n_out == 4, left=int(rng.integers(0, 5)))
    else:
        p.poseidon16(a, c, dst, permute=True)


def _ef(b, n):
    """an operand addressing n consecutive defined extension elements"""
    if b.rng.random() < 0.5 or not b.efs or n > 1:
        return ("data", int(b.rng.integers(0, b.data_words - 5 * n)))
    return ("fp", int(b.rng.choice(b.efs)))


def op_extension(b):
    p, rng = b.p, b.rng
    op = str(rng.choice(["add", "mul", "poly_eq"]))
    be = bool(rng.random() < 0.4)
    size = int(rng.integers(1, 7))
    a, c = _operands(b, _ef(b, size), _ef(b, size))
    outside = rng.random() < 0.2
    dst = None
    if outside:
        oc = b.out_ptr(5)
        if oc is not None:
            dst = M(oc)
    if dst is None:
        o = b.alloc(5)
        dst = FP(o)
        b.efs.append(o)
        b.words.append(o)
    p.extension_op(op, a, c, dst, size=size, is_be=be)


def op_extension_solve(b):
    """size-1 add / mul with one unknown operand and a known result (exec.rs:29-94), incl. the copy_5 forms"""
    p, rng = b.p, b.rng
    if not b.efs:
        return
    known, res = int(rng.choice(b.efs)), int(rng.choice(b.efs))
    new = b.alloc(5)
    op = str(rng.choice(["add", "mul"]))
    if rng.random() < 0.5:
        p.extension_op(op, FP(new), FP(known), FP(res))
    else:
        p.extension_op(op, FP(known), FP(new), FP(res))
    b.efs.append(new)
    b.words.append(new)


def op_hint_inverse(b):
    p = b.p
    x, sq, nz, inv, one = b.word(), b.alloc(), b.alloc(), b.alloc(), b.alloc()
    p.mul(M(x), M(x), M(sq))
    p.add(M(sq), K(1), M(nz))
    p.hint_inverse(M(nz), inv)
    p.mul(M(nz), M(inv), M(one))
    p.add(M(one), K(0), K(1))
    b.words += [nz, inv]


def op_hint_decompose(b):
    p, rng = b.p, b.rng
    kind = int(rng.integers(0, 3))
    x = b.word()
    if kind == 0:
        chunk = int(rng.choice([1, 2, 3, 4, 6, 8, 12]))
        n = 24 // chunk
        dst = b.alloc(2 * n)
        src = b.alloc(2)
        p.add(M(x), K(0), M(src))
        p.add(M(b.word()), K(3), M(src + 1))
        p.hint_decompose_bits_xmss(FP(dst), FP(src), K(2), K(chunk))
        p.add(M(dst), K(0), M(b.alloc()))
        b.words += [dst, dst + 2 * n - 1]
    elif kind == 1:
        chunk = int(rng.choice([2, 4, 6, 12]))
        n = 24 // chunk
        dst = b.alloc(n)
        p.hint_decompose_bits_merkle_whir(FP(dst), M(x), K(chunk))
        p.add(M(dst + n - 1), K(0), M(b.alloc()))
        b.words += [dst, dst + n - 1]
    else:
        nb = int(rng.integers(1, 32))
        dst = b.alloc(nb)
        p.hint_decompose_bits(M(x), FP(dst), K(nb))
        p.add(M(dst), K(1), M(b.alloc()))
        b.words.append(dst)


def op_hint_misc(b):
    p, rng = b.p, b.rng
    r = int(rng.integers(0, 3))
    if r == 0:
        res = b.alloc()
        p.hint_log2_ceil(M(int(rng.choice(b.small))), M(res))
        p.add(M(res), K(0), M(b.alloc()))
        b.words.append(res)
    elif r == 1:
        x = b.word()
        p.hint_debug_assert(M(x), M(x), "==")
        p.hint_debug_assert(M(int(rng.choice(b.small))), K(b.n_iter + 100), "<")
        p.add(M(x), K(0), M(b.alloc()))
    else:
        name = f"w{b.tag}_{len(b.hints)}"
        n = int(rng.integers(1, 12))
        dst = b.alloc(n)
        p.hint_witness(name, dst)
        p.add(M(dst), K(0), M(b.alloc()))
        b.hints[name] = n
        b.words += [dst, dst + n - 1]
        if n >= 8:
            b.blocks.append(dst)
        if n >= 5:
            b.efs.append(dst)


def op_read_digest(b):
    """a word of a digest main's head computes: a pending cell when the calls are deferred (forces the chain, or hands the batch back)"""
    if not b.has_digest:
        return
    new = b.alloc()
    b.p.deref(7, int(b.rng.integers(0, 8)), M(new))
    b.words.append(new)


OPS = [(op_arith, 5), (op_solve, 2), (op_assert, 1), (op_load_data, 3), (op_store_out, 3), (op_frame_pointer_store_load, 2), (op_range_check, 2),
       (op_branch, 2), (op_table, 1), (op_poseidon, 5), (op_extension, 5), (op_extension_solve, 2), (op_hint_inverse, 1), (op_hint_decompose, 2),
       (op_hint_misc, 2), (op_read_digest, 1)]


SPARE = 12


def emit_loop(p, rng, tag, n_iter, data_words, has_digest, fault, parallel=True, chain=False):
    """a loop function `loop<tag>`; returns (frame size, inline hint streams {name: words per entry})"""
    n_args = 6                                     # i, end, OUT, DATA, PERM, DIG
    b = Body(p, rng, n_args, tag, n_iter, fault)
    b.data_words, b.has_digest = data_words, has_digest
    if parallel:
        p.hint_parallel_batch_start(n_args, M(3))
    p.label("loop" + tag)
    d, inv, nz, omnz = 2 + n_args, 3 + n_args, 4 + n_args, 5 + n_args
    p.add(M(d), M(3), M(2))
    p.hint_inverse(M(d), inv)
    p.mul(M(d), M(inv), M(nz))
    p.add(M(omnz), M(nz), K(1))
    p.mul(M(omnz), M(d), K(0))
    p.jump(M(nz), K(Label("body" + tag)), FP(0))
    p.jump(K(1), M(0), M(1))
    p.label("body" + tag)
    spare = b.alloc()                              # cell 12 of every frame: main defines it in the first frame of a loop
    assert spare == SPARE
    if chain:                                      # ... and with `chain` every iteration defines the NEXT frame's from its own: a write into
        b.words.append(spare)                      # another frame that the other frame READS — sequentially fine, impossible inside a batch
    blk = b.alloc(16)
    p.hint_witness("blk" + tag, blk)               # 16 hinted words per iteration
    b.hints["blk" + tag] = 16
    p.add(M(blk), K(0), M(b.alloc()))
    b.blocks += [blk, blk + 8]
    b.efs += [blk, blk + 5, blk + 10]
    b.words += [blk, blk + 15]
    fns, wts = zip(*OPS)
    wts = np.asarray(wts, dtype=float) / sum(wts)
    n_ops = int(rng.integers(3, 13))
    fault_at = int(rng.integers(0, n_ops)) if fault else -1
    for k in range(n_ops):
        if k == fault_at:
            emit_fault(b, fault)
        fns[int(rng.choice(len(fns), p=wts))](b)
    nxt, ip1 = b.alloc(), b.alloc()
    frame = b.top
    p.hint_request_memory(nxt, K(Label("@frame" + tag)))
    p.deref(nxt, 0, M(0))
    p.deref(nxt, 1, M(1))
    p.add(M(2), K(1), M(ip1))
    p.deref(nxt, 2, M(ip1))
    for a in range(3, 2 + n_args):
        p.deref(nxt, a, M(a))
    if chain:
        t = b.alloc()
        frame = b.top
        p.mul(M(spare), K(3), M(t))
        p.deref(nxt, SPARE, M(t))
    p.jump(K(1), K(Label("loop" + tag)), M(nxt))
    for tab, after, y, consts in b.tables:
        p.label(tab)
        for cst in consts:
            p.add(K(0), K(cst), M(y))
            p.jump(K(1), K(Label(after)), FP(0))
    return frame, b.hints


def emit_fault(b, fault):
    p = b.p
    if fault == "assert_in_one_iteration":         # data[5 i] is zero for one i only: its inverse hint gives 0 and 0 * 0 != 1
        i5, ptr, x, inv = b.alloc(), b.alloc(), b.alloc(), b.alloc()
        p.mul(M(2), K(5), M(i5))
        p.add(M(5), M(i5), M(ptr))
        p.deref(ptr, 0, M(x))
        p.hint_inverse(M(x), inv)
        p.mul(M(x), M(inv), K(1))
    elif fault == "conflicting_writes":             # every iteration writes i into ONE cell of main's OUT array
        p.deref(4, OUTW // 2 - 1, M(2))
        b.out_end -= 1
    elif fault == "undefined_read":                 # a read of a cell of main's frame that nothing ever defines (cell 1 = the caller's fp)
        new, y = b.alloc(), b.alloc()
        p.deref(1, 140, M(new))
        p.add(M(new), K(1), M(y))
    elif fault == "extension_check":                # ones * blk == blk + (i != 0): holds in iteration 0 only
        blk = b.blocks[0]
        p.extension_op("mul", FP(blk), FP(blk + 5), FP(blk + 10))


FAULTS = ["assert_in_one_iteration", "conflicting_writes", "undefined_read", "extension_check"]


def gen(seed, device=False):
    """-> (Bytecode, public input, Witness, description)"""
    rng = np.random.default_rng(seed)
    n1 = int(rng.integers(33, 49)) if device else int(rng.integers(2, 10))
    two = rng.random() < 0.3
    n2 = min(n1, int(rng.integers(33, 41)) if device and rng.random() < 0.5 else int(rng.integers(2, 9))) if two else 0
    fault = str(rng.choice(FAULTS)) if rng.random() < 0.2 else None
    head_chain = int(rng.integers(1, 5)) if rng.random() < 0.6 else 0
    data_words = max(int(rng.integers(64, 200)), 5 * n1 + 8)
    p = Program()
    N, OUT, DATA, PERM, LF, N2, LF2, DIG = 0, 1, 2, 3, 4, 5, 6, 7
    p.add(K(0), K(0), M(20))
    p.hint_witness("n", N)
    p.hint_witness("n2", N2)
    sz = 21
    p.mul(M(N), K(OUTW), M(sz))
    p.hint_request_memory(OUT, M(sz))
    p.hint_request_memory(DATA, K(data_words))
    p.hint_witness("data", DATA, indirect=True)
    p.hint_request_memory(PERM, M(N))
    p.hint_witness("perm", PERM, indirect=True)
    p.add(M(N), K(0), M(22))
    # the head: a chain of compressions nothing reads before the loop (deferred when the runner records calls)
    p.hint_witness("head_block", HEAD_BLOCK)
    p.add(K(0), FP(HEAD_DIG + 8 * max(0, head_chain - 1)), M(DIG))
    for k in range(head_chain):
        src = FP(HEAD_BLOCK) if k == 0 else FP(HEAD_DIG + 8 * (k - 1))
        p.poseidon16(src, FP(HEAD_BLOCK + 8), FP(HEAD_DIG + 8 * k))
    for k in range(5):
        p.add(K(0), K(1 if k == 0 else 0), M(150 + k))
    if head_chain and rng.random() < 0.5:          # a deferred ExtensionOp over the last digest: copy_5 onto fresh cells
        p.extension_op("mul", FP(HEAD_DIG + 8 * (head_chain - 1)), FP(150), FP(120))

    def call(lf, label, ret, n_cell, frame):
        p.hint_request_memory(lf, K(Label(frame)))
        p.deref(lf, 0, K(Label(ret)))
        p.deref(lf, 1, FP(0))
        p.deref(lf, 2, K(0))
        p.deref(lf, 3, M(n_cell))
        for k, c in enumerate((OUT, DATA, PERM, DIG)):
            p.deref(lf, 4 + k, M(c))
        p.deref(lf, SPARE, K(int(rng.integers(1, P))))
        p.jump(K(1), K(Label(label)), M(lf))
        p.label(ret)

    call(LF, "loop_a", "after_a", N, "@frame_a")
    if two:
        call(LF2, "loop_b", "after_b", N2, "@frame_b")
    # the tail: reads what the loop's deferred writes defined and a cell of a loop frame
    if rng.random() < 0.5 and not fault:
        t = 30
        p.deref(LF, 2, M(t))                       # iteration 0's counter, through the frame pointer
        p.add(M(t), K(5), M(t + 1))
    p.return_from_main(23)
    p.starting_frame_memory = MAIN_FRAME
    fa, ha = emit_loop(p, rng, "_a", n1, data_words, head_chain > 0, fault)
    p.labels["@frame_a"] = fa
    hb = {}
    if two:
        # `chain`: loop b's iterations depend on each other through their frames; the reference runs a second parallel loop sequentially
        # (its runner arms one batch per run), this library batches it and falls back to the literal run when a segment fails
        fb, hb = emit_loop(p, rng, "_b", n2, data_words, head_chain > 0, None, parallel=bool(rng.random() < 0.7), chain=bool(rng.random() < 0.3))
        p.labels["@frame_b"] = fb
    bc = p.finalize()
    data = rng.integers(1, P, size=data_words)
    if fault == "assert_in_one_iteration":
        data[5 * int(rng.integers(1, n1))] = 0
    hints = {"n": [mont([n1])], "n2": [mont([max(n2, 1)])], "data": [mont(data)], "perm": [mont(rng.permutation(n1))],
             "head_block": [mont(rng.integers(0, P, size=16))]}
    for names, n in ((ha, n1), (hb, n2)):
        for name, words in names.items():
            hints[name] = [mont(rng.integers(0, P, size=words)) for _ in range(n)]
    for name in bc.names:
        hints.setdefault(name, [])
    pi = mont(rng.integers(0, P, size=8))
    return bc, pi, Witness(bc, 0, hints), dict(n1=n1, n2=n2, fault=fault, head_chain=head_chain, instructions=len(p.rows))


def kind_of(msg):
    """the RunnerError kind of an error text (library: 'lmh_...: pc N: Kind...', oracle: 'oracle VM: Kind...')"""
    m = re.search(r"(?:pc \d+: |oracle VM: )(?:ParallelSegmentFailed\(\d+, )?(?:Panic: )?([A-Za-z]+)", msg)
    return m.group(1) if m else msg


def run_all(seed, orc, ob, ctx=None, device=False):
    """-> None when every path agrees, else a description of the disagreement"""
    import leanmultisig_amd as lm
    bc, pi, w, meta = gen(seed, device)
    results = []
    modes = [("host eager", "0", None), ("host deferred", "1", None)] + ([("device", None, ctx)] if ctx is not None else [])
    for name, lazy, c in modes:
        if lazy is None:
            os.environ.pop("LM_VM_LAZY", None)
        else:
            os.environ["LM_VM_LAZY"] = lazy
        try:
            ex = vm.execute(bc, pi, w, n_threads=2, ctx=c)
            results.append((name, ex, None))
        except lm.LmError as e:
            results.append((name, None, str(e)))
    os.environ.pop("LM_VM_LAZY", None)
    try:
        run, oerr = ob.VmRun(orc, bc, pi, w), None
    except RuntimeError as e:
        run, oerr = None, str(e)
    for name, ex, err in results:
        if (err is None) != (oerr is None):
            return f"seed {seed} {meta}: {name} {'failed: ' + err if err else 'succeeded'}, the oracle {'failed: ' + oerr if oerr else 'succeeded'}"
        if err is not None:
            if kind_of(err) != kind_of(oerr):
                return f"seed {seed} {meta}: {name}: {err} / oracle: {oerr}"
            if err != results[0][2]:
                return f"seed {seed} {meta}: error texts differ: {name}: {err} / {results[0][0]}: {results[0][2]}"
            continue
        if not (ex.n_cycles == run.pcs.size and ex.memory_len == run.memory.size and np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)):
            return f"seed {seed} {meta}: {name}: the cycle log differs from the oracle's ({ex.n_cycles} / {run.pcs.size} cycles, memory {ex.memory_len} / {run.memory.size})"
        if not (np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)):
            bad = np.nonzero((ex.memory() != run.memory) | (ex.memory_defined() != run.defined))[0]
            return f"seed {seed} {meta}: {name}: memory differs from the oracle's at {bad[:6]}"
        if not (ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows):
            return f"seed {seed} {meta}: {name}: counts differ from the oracle's"
        if ex is not results[0][1] and results[0][1] is not None:
            e0 = results[0][1]
            if not (np.array_equal(ex.poseidon_calls(), e0.poseidon_calls()) and np.array_equal(ex.extension_rows(), e0.extension_rows())):
                return f"seed {seed} {meta}: {name}: precompile records differ from {results[0][0]}'s"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=1000)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--device", action="store_true")
    args = ap.parse_args()
    from tests import oracle_binding as ob
    orc = ob.load()
    ctx = None
    if args.device:
        import leanmultisig_amd as lm
        ctx = lm.Context(0)
    t0 = time.time()
    bad, faults, on_device = [], 0, 0
    for s in range(args.first, args.first + args.seeds):
        r = run_all(s, orc, ob, ctx, args.device)
        if r:
            bad.append(r)
            print(r, flush=True)
    print(f"{args.seeds} programs from seed {args.first}: {len(bad)} disagreements, {time.time() - t0:.1f} s"
          + (" (host eager, host deferred, device, oracle)" if args.device else " (host eager, host deferred, oracle)"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
