"""leanVM runner (lmh_execute_bytecode, host C++) against the oracle's restatement (oracle/vm_oracle.hpp) and against
hand-computed expectations: every instruction kind and operand-unknown case of execute_instruction (lean_vm/src/isa/instruction.rs:
146-246), every hint (isa/hint.rs), both precompiles with all their variants, the deref-hint resolution, parallel loop batches on
1 and several threads, RunnerErrors.  CPU only: the runner is host code; the C-ABI library is loaded, no device call is made."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from leanmultisig_amd import vm
from leanmultisig_amd.vm import FP, K, M, Label, Program, Witness, execute, from_monty, to_monty
from leanmultisig_amd.programs import xmss_aggregate as xa
from tests import oracle_binding as ob
from tests import synth_witness

P = 0x7F000001
PI = to_monty(np.arange(1, 9)).astype(np.uint32)   # public input: memory[0..8] = 1..8; main fp = 10 with an empty preamble


def mont(x):
    return to_monty(x).astype(np.uint32)


def both(orc, p, hints=None, pi=PI, preamble=0, n_threads=1, log_size=None):
    """run on the library and on the oracle; assert identical logs; -> (Execution, canonical memory, defined mask)"""
    bc = p.finalize(log_size)
    w = Witness(bc, preamble, hints or {})
    ex = execute(bc, pi, w, n_threads=n_threads)
    run = ob.VmRun(orc, bc, pi, w)
    assert np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows
    assert (ex.public_memory_size, ex.runtime_memory_size) == (run.public_memory_size, run.runtime_memory_size)
    return ex, from_monty(ex.memory()), ex.memory_defined(), run, bc


def fails(orc, p, match, hints=None, pi=PI):
    bc = p.finalize()
    w = Witness(bc, 0, hints or {})
    with pytest.raises(lm.LmError, match=match):
        execute(bc, pi, w, n_threads=1)
    with pytest.raises(RuntimeError):
        ob.VmRun(orc, bc, pi, w)


FP0 = 10  # next_multiple_of(8 + 0, 5)


@pytest.fixture(autouse=True, params=["eager", "deferred"])
def poseidon_mode(request, monkeypatch):
    """every test of this module runs twice: with the sequential runner's Poseidon calls executed at once, and recorded and executed
    on first use (csrc/host/lm_vm.cpp: MemBuf — the mode of a run with a device context; LM_VM_LAZY forces it on the host)"""
    monkeypatch.setenv("LM_VM_LAZY", "1" if request.param == "deferred" else "0")
    return request.param


def test_computation_all_unknown_cases(orc):
    p = Program()
    p.starting_frame_memory = 32
    p.add(K(0), K(7), M(0))            # res unknown: 0 + 7
    p.mul(M(0), K(6), M(1))            # 42
    p.add(M(2), M(0), M(1))            # a unknown: 42 - 7 = 35
    p.add(M(0), M(3), M(1))            # c unknown: 35
    p.mul(M(4), M(0), M(1))            # a unknown: 42 / 7 = 6
    p.mul(M(0), M(5), M(1))            # c unknown: 6
    p.add(M(2), M(0), M(1))            # all known: 35 + 7 == 42
    p.mul(M(4), M(5), K(36))           # all known, constant result
    p.add(K(0), FP(3), M(6))           # fp-relative operand: fp + 3
    p.mul(K(2), FP(0), M(7))           # 2 * fp
    p.add(K(P - 1), K(5), M(8))        # wrap-around: -1 + 5 = 4
    p.return_from_main(9)
    ex, mem, defd, _, bc = both(orc, p)
    assert list(mem[FP0:FP0 + 10]) == [7, 42, 35, 35, 6, 6, FP0 + 3, 2 * FP0, 4, 0]
    assert ex.counts == dict(add=6 + 1, mul=5, deref=0, jump=1)
    assert ex.n_cycles == 14 and ex.pcs()[-1] == bc.ending_pc and ex.fps()[-1] == 0


def test_deref_load_store_and_range_check(orc):
    p = Program()
    p.starting_frame_memory = 32
    p.add(K(0), K(3), M(0))            # pointer to the public input word 3 (value 4)
    p.deref(0, 2, M(1))                # load m[3 + 2] = 6
    p.add(K(0), FP(20), M(2))          # pointer into the own frame
    p.deref(2, 1, K(99))               # store m[fp + 21] = 99
    p.deref(2, 2, M(1))                # store m[fp + 22] = m[fp + 1] = 6
    p.deref(2, 3, FP(5))               # store fp + 5
    p.add(K(0), K(5), M(3))            # range check 5 <= 9 (b_compile_intermediate.rs:673-752)
    p.range_check(3, K(9), 4)
    p.add(K(0), K(2000), M(8))         # an address nobody writes: the deref stays undefined until resolve_deref_hints zero-fills it
    p.hint_deref(8, 9)
    p.deref(8, 0, M(9))
    p.return_from_main(10)
    ex, mem, defd, _, _ = both(orc, p)
    assert mem[FP0 + 1] == 6 and list(mem[FP0 + 21:FP0 + 24]) == [99, 6, FP0 + 5]
    assert mem[FP0 + 4] == 6                       # m[5]: the range-check deref reads the public input
    assert mem[FP0 + 5] == 4 and mem[FP0 + 6] == 5  # complement 9 - 5 = 4, m[4] = 5
    assert mem[FP0 + 9] == 0 and defd[FP0 + 9] == 1


def test_jump_call_and_return(orc):
    p = Program()
    p.starting_frame_memory = 8
    p.add(K(0), K(0), M(0))
    p.jump(M(0), K(Label("never")), FP(0))         # condition 0: falls through
    p.hint_request_memory(1, K(6))                  # callee frame
    p.deref(1, 0, K(Label("back")))                 # return pc
    p.deref(1, 1, FP(0))                            # saved fp
    p.deref(1, 2, K(21))                            # argument
    p.jump(K(1), K(Label("double")), M(1))
    p.label("back")
    p.deref(1, 3, M(2))                             # result
    p.return_from_main(3)
    p.label("never")
    p.panic()
    p.label("double")                               # frame: [ret, fp, x, 2x]
    p.add(M(2), M(2), M(3))
    p.jump(K(1), M(0), M(1))
    ex, mem, _, _, _ = both(orc, p)
    assert mem[FP0 + 2] == 42 and mem[FP0 + 1] == FP0 + 8
    assert list(ex.fps()[5:8]) == [FP0, FP0 + 8, FP0 + 8]


def test_poseidon_variants(orc):
    rng = np.random.default_rng(1)
    blob = ob.rand_field(rng, 32)
    p = Program()
    p.starting_frame_memory = 128
    p.hint_witness("blob", 0)
    p.poseidon16(FP(0), FP(8), FP(40))                          # compress: both operands fp-relative
    p.add(K(0), FP(0), M(32))
    p.add(K(0), FP(16), M(33))
    p.poseidon16(M(32), M(33), FP(48), half=True)               # pointers in cells
    p.poseidon16(M(32), K(0), FP(56), left=4)                   # hardcoded left: m[4..8] | m[a..a+4], right = the public input
    p.poseidon16(FP(8), FP(16), FP(64), half=True, left=2)
    p.poseidon16(FP(0), FP(16), FP(72), permute=True)
    p.return_from_main(100)
    ex, mem, defd, run, _ = both(orc, p, {"blob": [blob]})
    m = ex.memory()
    f = lambda a, n: m[FP0 + a:FP0 + a + n]  # noqa: E731
    pub = PI
    exp0 = orc.poseidon16_compress(np.concatenate([f(0, 8), f(8, 8)]))[0]
    assert np.array_equal(f(40, 8), exp0[:8])
    exp1 = orc.poseidon16_compress(np.concatenate([f(0, 8), f(16, 8)]))[0]
    assert np.array_equal(f(48, 4), exp1[:4]) and not defd[FP0 + 52]
    exp2 = orc.poseidon16_compress(np.concatenate([pub[4:8], f(0, 4), pub]))[0]
    assert np.array_equal(f(56, 8), exp2[:8])
    exp3 = orc.poseidon16_compress(np.concatenate([pub[2:6], f(8, 4), f(16, 8)]))[0]
    assert np.array_equal(f(64, 4), exp3[:4])
    exp4 = orc.poseidon16_permute(np.concatenate([f(0, 8), f(16, 8)]))[0]
    assert np.array_equal(f(72, 16), exp4)
    calls = ex.poseidon_calls()
    assert calls.shape == (5, 9)
    assert list(calls[2]) == [FP0, 0, FP0 + 56, 0, 1, 4, 4, FP0, 0] and list(calls[4][3:]) == [0, 0, 0, FP0, FP0 + 4, 1]


def test_poseidon_chain_cells_written_and_read_around_the_calls(orc):
    """a hash chain nobody reads for a while (deferred mode: pending calls), then every way of meeting a pending cell: a read that has
    to execute the chain, writes of the right and of a wrong value to an output cell — before and after the call —, a call whose output
    lands on a defined cell, a call on an undefined input, an extension operation over pending cells"""
    rng = np.random.default_rng(7)
    blob = ob.rand_field(rng, 16)

    def chain(p, n=6):
        p.hint_witness("blob", 0)
        p.poseidon16(FP(0), FP(8), FP(40))
        for k in range(1, n):
            p.poseidon16(FP(40 + 8 * (k - 1)), FP(8), FP(40 + 8 * k), half=(k == n - 1))
        p.poseidon16(FP(40), FP(48), FP(100), permute=True)

    def expected():
        h = [orc.poseidon16_compress(np.concatenate([blob[:8], blob[8:]]))[0][:8]]
        for k in range(1, 6):
            h.append(orc.poseidon16_compress(np.concatenate([h[-1], blob[8:]]))[0][:8])
        return h, orc.poseidon16_permute(np.concatenate([h[0], h[1]]))[0]

    h, perm = expected()
    hc, permc = [from_monty(x) for x in h], from_monty(perm)

    def prog(build, before=None):
        p = Program()
        p.starting_frame_memory = 160
        if before:
            before(p)
        chain(p)
        build(p)
        p.return_from_main(150)
        return p

    # nothing reads the chain: it is executed when the run ends
    ex, mem, defd, _, _ = both(orc, prog(lambda p: None), {"blob": [blob]})
    assert np.array_equal(mem[FP0 + 40:FP0 + 48], hc[0]) and np.array_equal(mem[FP0 + 80:FP0 + 84], hc[5][:4]) and not defd[FP0 + 84]
    assert np.array_equal(mem[FP0 + 100:FP0 + 116], permc)
    # a read in the middle of the chain, an unknown operand solved from a pending cell, the right value written onto pending cells
    def reads(p):
        p.add(M(64), K(1), M(120))                       # h[3][0] + 1
        p.add(M(121), M(82), K(5))                       # a unknown: 5 - h[5][2]
        p.add(K(0), K(int(hc[2][3])), M(59))             # the value the pending call will write
        p.add(K(0), K(int(permc[15])), M(115))
        p.extension_op("add", FP(72), FP(100), FP(130))  # h[4][0..5] + perm[0..5]
    ex, mem, _, _, _ = both(orc, prog(reads), {"blob": [blob]})
    assert mem[FP0 + 120] == (int(hc[3][0]) + 1) % P and mem[FP0 + 121] == (5 - int(hc[5][2])) % P
    assert list(mem[FP0 + 130:FP0 + 135]) == [(int(a) + int(b)) % P for a, b in zip(hc[4][:5], permc[:5])]
    # an output cell defined BEFORE the call: the right value passes, a wrong one is the call's MemoryAlreadySet
    both(orc, prog(lambda p: None, before=lambda p: p.add(K(0), K(int(hc[1][7])), M(55))), {"blob": [blob]})
    fails(orc, prog(lambda p: None, before=lambda p: p.add(K(0), K((int(hc[1][7]) + 1) % P), M(55))), "MemoryAlreadySet", {"blob": [blob]})
    # a wrong value written AFTER the call
    fails(orc, prog(lambda p: p.add(K(0), K((int(hc[4][0]) + 1) % P), M(72))), "NotEqual", {"blob": [blob]})
    fails(orc, prog(lambda p: (p.add(K(0), FP(0), M(122)), p.deref(122, 109, K((int(permc[9]) + 1) % P)))), "MemoryAlreadySet", {"blob": [blob]})
    both(orc, prog(lambda p: (p.add(K(0), FP(0), M(122)), p.deref(122, 109, K(int(permc[9]))))), {"blob": [blob]})
    # checks over pending cells (deferred mode: recorded, executed with the chain): copy_5 onto the right / a wrong expected value, a dot
    # product against a defined result; a failing check followed by another error reports the check (the first error of the program),
    # a passing one followed by an error reports that error
    def expect(p, value5, at=140):
        for k in range(5):
            p.add(K(0), K(int(value5[k]) % P), M(at + k))
    one = lambda p: (p.add(K(0), K(1), M(135)), [p.add(K(0), K(0), M(136 + k)) for k in range(4)])  # noqa: E731
    def copy_check(delta, then=None):
        def build(p):
            one(p)
            expect(p, [int(hc[4][0]) + delta] + list(hc[4][1:5]))
            p.extension_op("mul", FP(72), FP(135), FP(140))
            if then:
                then(p)
        return build
    ex, _, _, _, _ = both(orc, prog(copy_check(0)), {"blob": [blob]})
    assert ex.n_extension_rows == 1
    fails(orc, prog(copy_check(1)), "NotEqual", {"blob": [blob]})
    fails(orc, prog(copy_check(1, then=lambda p: p.add(M(0), K(1), M(0)))), "NotEqual", {"blob": [blob]})
    fails(orc, prog(copy_check(0, then=lambda p: p.jump(K(1), K(100000), FP(0)))), "PCOutOfBounds", {"blob": [blob]})
    dot = sum(int(a) * int(b) for a, b in zip(hc[2][:3], hc[3][:3])) % P  # base-by-extension dot product of size 1: h[2][0] * (h[3][0..5])
    def dot_check(delta):
        def build(p):
            prod = [(int(hc[2][0]) * int(x)) % P for x in hc[3][:5]]
            expect(p, [prod[0] + delta] + prod[1:])
            p.extension_op("mul", FP(56), FP(64), FP(140), is_be=True)
        return build
    both(orc, prog(dot_check(0)), {"blob": [blob]})
    fails(orc, prog(dot_check(1)), "InvalidExtensionOp|NotEqual|MemoryAlreadySet", {"blob": [blob]})
    # a call over an undefined input behind pending ones; a call whose output runs into a pending cell of another call
    fails(orc, prog(lambda p: p.poseidon16(FP(80), FP(8), FP(140))), "UndefinedMemory", {"blob": [blob]})   # h[5] is a half output
    fails(orc, prog(lambda p: p.poseidon16(FP(40), FP(8), FP(36))), "MemoryAlreadySet", {"blob": [blob]})  # 36..44 overlaps h[0]


@pytest.mark.parametrize("op,is_be,size", [("add", False, 1), ("mul", False, 1), ("mul", False, 4), ("mul", True, 3), ("poly_eq", False, 3),
                                           ("poly_eq", True, 2), ("add", True, 2), ("add", False, 3)])
def test_extension_op_modes(orc, op, is_be, size):
    rng = np.random.default_rng(7)
    a = ob.rand_field(rng, size if is_be else 5 * size)
    b = ob.rand_field(rng, 5 * size)
    p = Program()
    p.starting_frame_memory = 128
    p.hint_witness("a", 0)
    p.hint_witness("b", 40)
    p.extension_op(op, FP(0), FP(40), FP(80), size=size, is_be=is_be)
    p.return_from_main(100)
    ex, mem, _, run, _ = both(orc, p, {"a": [a], "b": [b]})
    rows = ex.extension_rows()
    assert rows.shape == (size, 24) and list(rows[:, 5]) == list(range(size, 0, -1)) and rows[0, 1] == 1
    # the oracle's table rows carry the same values (checked column by column in the GPU test); here: the result cell
    av = np.zeros((size, 5), dtype=np.uint32)
    if is_be:
        av[:, 0] = a
    else:
        av = a.reshape(size, 5)
    bv = b.reshape(size, 5)
    one = np.array([lm.capi.P and 0x01FFFFFE, 0, 0, 0, 0], dtype=np.uint32)
    if op == "add":
        elems = [synth_witness.ef_add(av[i], bv[i]) if hasattr(synth_witness, "ef_add") else (av[i].astype(np.uint64) + bv[i]) % P for i in range(size)]
        acc = np.zeros(5, dtype=np.uint64)
        for e in elems:
            acc = (acc + np.asarray(e, dtype=np.uint64)) % P
    elif op == "mul":
        acc = np.zeros(5, dtype=np.uint64)
        for i in range(size):
            acc = (acc + orc.ef_mul(av[i], bv[i])) % P
    else:
        acc = one.astype(np.uint64)
        for i in range(size):
            ab = orc.ef_mul(av[i], bv[i]).astype(np.uint64)
            e = (2 * ab + 2 * P - av[i] - bv[i] + one) % P
            acc = orc.ef_mul(acc.astype(np.uint32), e.astype(np.uint32)).astype(np.uint64)
    assert np.array_equal(ex.memory()[FP0 + 80:FP0 + 85], acc.astype(np.uint32))


def test_extension_op_solves_unknowns_and_copies(orc):
    rng = np.random.default_rng(9)
    a, b = ob.rand_field(rng, 5), ob.rand_field(rng, 5)
    p = Program()
    p.starting_frame_memory = 128
    p.hint_witness("a", 0)
    p.hint_witness("b", 8)
    for k, v in enumerate([1, 0, 0, 0, 0]):
        p.add(K(0), K(v), M(16 + k))                               # ONE in the extension field
    p.extension_op("mul", FP(0), FP(8), FP(24))                    # c = a * b
    p.extension_op("mul", FP(32), FP(8), FP(24))                   # A unknown: c / b
    p.extension_op("mul", FP(0), FP(40), FP(24))                   # B unknown: c / a
    p.extension_op("add", FP(0), FP(8), FP(48))                    # s = a + b
    p.extension_op("add", FP(56), FP(8), FP(48))                   # A unknown: s - b
    p.extension_op("mul", FP(0), FP(16), FP(64))                   # copy_5: dst <- src
    p.extension_op("mul", FP(72), FP(16), FP(0))                   # copy_5 backwards: src <- dst
    p.extension_op("mul", FP(80), FP(16), FP(88))                  # both unknown: zeros
    p.extension_op("mul", FP(16), FP(96), FP(8))                   # a == ONE: b <- res
    p.return_from_main(120)
    ex, _, defd, _, _ = both(orc, p, {"a": [a], "b": [b]})
    m = ex.memory()
    g = lambda o: m[FP0 + o:FP0 + o + 5]  # noqa: E731
    assert np.array_equal(g(32), a) and np.array_equal(g(40), b) and np.array_equal(g(56), a)
    assert np.array_equal(g(64), a) and np.array_equal(g(72), a) and not g(80).any() and not g(88).any() and defd[FP0 + 80]
    assert np.array_equal(g(96), b)
    assert ex.n_extension_rows == 9


def test_hints(orc):
    p = Program()
    p.starting_frame_memory = 128
    p.add(K(0), K(1234567), M(0))
    p.hint_inverse(M(0), 1)
    p.mul(M(0), M(1), K(1))
    p.hint_inverse(K(0), 2)                                         # inverse(0) = 0
    p.add(M(2), K(0), K(0))
    p.hint_request_memory(3, K(10))
    p.hint_request_memory(4, M(0))                                  # size from memory
    p.hint_request_memory(5, K(1))
    p.add(M(3), K(10), M(4))                                        # consecutive allocations
    p.add(M(4), K(1234567), M(5))
    p.add(K(0), K(0b1011_0110_1100_0011_1010_0101 + (5 << 24)), M(6))
    p.add(K(0), K(77), M(7))
    p.hint_decompose_bits_xmss(FP(8), FP(6), K(2), K(6))            # two elements, 6-bit chunks -> 8 cells
    p.hint_decompose_bits_merkle_whir(FP(16), M(6), K(8))           # one element, 8-bit chunks -> 3 cells
    p.hint_decompose_bits(M(7), FP(20), K(8))                       # big-endian bits of 77
    p.hint_less_than(M(7), K(78), M(28))
    p.hint_less_than(M(7), M(7), M(29))
    p.hint_log2_ceil(M(7), M(30))
    p.hint_log2_ceil(K(64), M(31))
    p.hint_debug_assert(M(7), K(78), "<")
    p.hint_debug_assert(M(7), K(100), "<=", preceds_runtime_inequality=True)
    p.hint_witness("data", 32)
    p.add(K(0), FP(40), M(36))
    p.hint_witness("data", 36, indirect=True)
    p.add(M(32), M(40), M(44))
    p.return_from_main(100)
    d = [mont([1, 2, 3]), mont([10, 20])]
    ex, mem, defd, _, _ = both(orc, p, {"data": d})
    x = 0b1011_0110_1100_0011_1010_0101
    assert mem[FP0 + 1] == pow(1234567, P - 2, P) and mem[FP0 + 2] == 0
    assert list(mem[FP0 + 8:FP0 + 12]) == [(x >> (6 * i)) & 63 for i in range(4)] and list(mem[FP0 + 12:FP0 + 16]) == [77 & 63, 1, 0, 0]
    assert list(mem[FP0 + 16:FP0 + 19]) == [(x >> (8 * i)) & 255 for i in range(3)]
    assert list(mem[FP0 + 20:FP0 + 28]) == [int(c) for c in format(77, "08b")]
    assert list(mem[FP0 + 28:FP0 + 32]) == [1, 0, 7, 6]
    assert list(mem[FP0 + 32:FP0 + 35]) == [1, 2, 3] and list(mem[FP0 + 40:FP0 + 42]) == [10, 20] and mem[FP0 + 44] == 11
    assert ex.runtime_memory_size == 10 + 1234567 + 1


def loop_program(body_hashes=2, with_store=True, parallel=True):
    """main calls a PARALLEL loop: iteration i hashes a hinted block with the public input, stores i into out[perm[i]] and chains nothing
    between iterations (the shape of main.py:161-167)."""
    p = Program()
    f = 0
    N, OUT, PERM, LF = 0, 1, 2, 3
    p.add(K(0), K(0), M(20))                                       # (operand_a = 0 first)
    p.hint_witness("n", N)
    p.hint_request_memory(OUT, M(N))
    p.hint_request_memory(PERM, M(N))
    p.hint_witness("perm", PERM, indirect=True)
    p.hint_request_memory(LF, K(Label("@frame")))
    p.deref(LF, 0, K(Label("after")))
    p.deref(LF, 1, FP(0))
    p.deref(LF, 2, K(0))
    p.deref(LF, 3, M(N))
    p.deref(LF, 4, M(OUT))
    p.deref(LF, 5, M(PERM))
    p.jump(K(1), K(Label("loop")), M(LF))
    p.label("after")
    p.return_from_main(21)
    p.starting_frame_memory = 32
    # frame: [ret, fp, i, end, out, perm | d, inv, nz, omnz, t, idx, o, blk(8), h(8 * body_hashes), next, ip1]
    I, END, OUTP, PERMP = 2, 3, 4, 5
    d, inv, nz, omnz, t, idx, o, blk = 6, 7, 8, 9, 10, 11, 12, 13
    h = 21
    nxt = h + 8 * body_hashes
    ip1 = nxt + 1
    frame = ip1 + 1
    if parallel:
        p.hint_parallel_batch_start(4, M(END))
    p.label("loop")
    p.add(M(d), M(END), M(I))
    p.hint_inverse(M(d), inv)
    p.mul(M(d), M(inv), M(nz))
    p.add(M(omnz), M(nz), K(1))
    p.mul(M(omnz), M(d), K(0))
    p.jump(M(nz), K(Label("body")), FP(0))
    p.jump(K(1), M(0), M(1))
    p.label("body")
    p.hint_witness("block", blk)
    p.poseidon16(FP(blk), FP(blk), FP(h))
    for k in range(1, body_hashes):
        p.poseidon16(FP(h + 8 * (k - 1)), FP(blk), FP(h + 8 * k))
    if with_store:
        p.add(M(PERMP), M(I), M(t))
        p.deref(t, 0, M(idx))
        p.add(M(OUTP), M(idx), M(o))
        p.deref(o, 0, M(I))                                        # a write OUTSIDE the iteration's frame: deferred in a segment
    p.hint_request_memory(nxt, K(frame))
    p.deref(nxt, 0, M(0))
    p.deref(nxt, 1, M(1))
    p.add(M(I), K(1), M(ip1))
    p.deref(nxt, 2, M(ip1))
    for a in (3, 4, 5):
        p.deref(nxt, a, M(a))
    p.jump(K(1), K(Label("loop")), M(nxt))
    p.labels["@frame"] = frame
    return p


@pytest.mark.parametrize("n,threads", [(1, 1), (2, 1), (9, 1), (9, 3), (64, 8)])
def test_parallel_batch(orc, n, threads):
    rng = np.random.default_rng(n)
    perm = rng.permutation(n)
    blocks = [ob.rand_field(rng, 8) for _ in range(n)]
    hints = {"n": [mont([n])], "perm": [mont(perm)], "block": blocks}
    ex, mem, defd, run, bc = both(orc, loop_program(), hints, n_threads=threads)
    out = int(mem[FP0 + 1])
    assert list(mem[out:out + n][perm]) == list(range(n))
    assert ex.n_poseidon_calls == 2 * n
    # iteration order in the log: the i-th pair of calls hashes the i-th block
    calls = ex.poseidon_calls()
    m = ex.memory()
    for i in (0, n - 1):
        assert np.array_equal(m[calls[2 * i, 0]:calls[2 * i, 0] + 8], blocks[i])
    # a sequential run (no ParallelBatchStart hint) produces the same log
    bc2 = loop_program(parallel=False).finalize()
    w2 = Witness(bc2, 0, hints)
    ex2 = execute(bc2, PI, w2)
    assert np.array_equal(ex2.pcs(), ex.pcs()) and np.array_equal(ex2.fps(), ex.fps())
    m1, m2 = ex.memory(), ex2.memory()            # the batch resizes the memory to the end of the last frame (runner.rs:404-407): undefined cells
    k = min(m1.size, m2.size)
    assert np.array_equal(m1[:k], m2[:k]) and not m1[k:].any() and not m2[k:].any()


def two_loops_program(body_hashes=2, with_store=True):
    """main calls TWO parallel loops in a row (the body of loop_program twice: n, then n2 iterations).  The reference batches the first
    and runs the second sequentially (one batch per run, runner.rs:120-198,262-298); this runner batches both and must leave the memory
    exactly as long as the sequential run does (lm_vm.cpp: trim_to_defined)."""
    q = Program()
    N, OUT, PERM, LF, N2, OUT2, PERM2, LF2 = 0, 1, 2, 3, 4, 5, 6, 7
    q.add(K(0), K(0), M(20))
    q.hint_witness("n", N)
    q.hint_request_memory(OUT, M(N))
    q.hint_request_memory(PERM, M(N))
    q.hint_witness("perm", PERM, indirect=True)
    q.hint_request_memory(LF, K(Label("@frame")))
    q.deref(LF, 0, K(Label("after")))
    q.deref(LF, 1, FP(0))
    q.deref(LF, 2, K(0))
    q.deref(LF, 3, M(N))
    q.deref(LF, 4, M(OUT))
    q.deref(LF, 5, M(PERM))
    q.jump(K(1), K(Label("loop")), M(LF))
    q.label("after")
    q.hint_witness("n2", N2)
    q.hint_request_memory(OUT2, M(N2))
    q.hint_request_memory(PERM2, M(N2))
    q.hint_witness("perm2", PERM2, indirect=True)
    q.hint_request_memory(LF2, K(Label("@frame")))
    q.deref(LF2, 0, K(Label("after2")))
    q.deref(LF2, 1, FP(0))
    q.deref(LF2, 2, K(0))
    q.deref(LF2, 3, M(N2))
    q.deref(LF2, 4, M(OUT2))
    q.deref(LF2, 5, M(PERM2))
    q.jump(K(1), K(Label("loopB")), M(LF2))
    q.label("after2")
    q.return_from_main(21)
    q.starting_frame_memory = 32
    I, END, OUTP, PERMP = 2, 3, 4, 5
    d, inv, nz, omnz, t, idx, o, blk = 6, 7, 8, 9, 10, 11, 12, 13
    h = 21
    nxt = h + 8 * body_hashes
    ip1 = nxt + 1
    frame = ip1 + 1
    for sfx in ("", "B"):   # two loop functions: a parallel loop is entered once per run (its armed batch belongs to one call frame)
        q.hint_parallel_batch_start(4, M(END))
        q.label("loop" + sfx)
        q.add(M(d), M(END), M(I))
        q.hint_inverse(M(d), inv)
        q.mul(M(d), M(inv), M(nz))
        q.add(M(omnz), M(nz), K(1))
        q.mul(M(omnz), M(d), K(0))
        q.jump(M(nz), K(Label("body" + sfx)), FP(0))
        q.jump(K(1), M(0), M(1))
        q.label("body" + sfx)
        q.hint_witness("block", blk)
        q.poseidon16(FP(blk), FP(blk), FP(h))
        for k in range(1, body_hashes):
            q.poseidon16(FP(h + 8 * (k - 1)), FP(blk), FP(h + 8 * k))
        if with_store:
            q.add(M(PERMP), M(I), M(t))
            q.deref(t, 0, M(idx))
            q.add(M(OUTP), M(idx), M(o))
            q.deref(o, 0, M(I))
        q.hint_request_memory(nxt, K(frame))
        q.deref(nxt, 0, M(0))
        q.deref(nxt, 1, M(1))
        q.add(M(I), K(1), M(ip1))
        q.deref(nxt, 2, M(ip1))
        for a in (3, 4, 5):
            q.deref(nxt, a, M(a))
        q.jump(K(1), K(Label("loop" + sfx)), M(nxt))
    q.labels["@frame"] = frame
    return q


@pytest.mark.parametrize("n,n2,with_store,threads", [(5, 7, True, 1), (5, 1, True, 2), (1, 1, True, 1), (9, 40, False, 4), (33, 2, False, 3), (4, 64, True, 8)])
def test_consecutive_parallel_batches_leave_the_sequential_memory_length(orc, n, n2, with_store, threads):
    """Round-5 advisor item: re-arming later parallel loops departs from the reference runner (which batches the first only); the
    ExecutionResult must stay the SEQUENTIAL one — pcs, fps, every memory cell and the memory LENGTH (it feeds log_memory and the public
    memory, hence the proof).  `both` compares all of them with the sequential oracle VM; the cases cover a second batch of one iteration
    (no growth beyond its call frame), one that is shorter / longer than the first, loops without stores outside their frames, and
    LM_VM_REARM=0's literal arming in a child run."""
    rng = np.random.default_rng(100 * n + n2)
    hints = {"n": [mont([n])], "perm": [mont(rng.permutation(n))], "n2": [mont([n2])], "perm2": [mont(rng.permutation(n2))],
             "block": [ob.rand_field(rng, 8) for _ in range(n + n2)]}
    ex, mem, defd, run, bc = both(orc, two_loops_program(with_store=with_store), hints, n_threads=threads)
    assert ex.memory_len == run.memory.size and ex.n_poseidon_calls == 2 * (n + n2)
    if with_store:
        out2 = int(mem[FP0 + 5])
        assert sorted(mem[out2:out2 + n2]) == list(range(n2))


def test_parallel_batch_conflicting_deferred_write_fails(orc):
    n = 6
    rng = np.random.default_rng(0)
    perm = np.array([0, 1, 2, 3, 4, 4])                             # two iterations store different values into one cell
    hints = {"n": [mont([n])], "perm": [mont(perm)], "block": [ob.rand_field(rng, 8) for _ in range(n)]}
    fails(orc, loop_program(), "MemoryAlreadySet", hints)


def test_runner_errors(orc):
    def prog(build):
        p = Program()
        p.starting_frame_memory = 16
        p.add(K(0), K(5), M(0))
        build(p)
        p.return_from_main(15)
        return p
    fails(orc, prog(lambda p: p.add(M(0), K(1), K(7))), "NotEqual")
    fails(orc, prog(lambda p: (p.add(K(0), FP(0), M(1)), p.deref(1, 0, K(6)))), "MemoryAlreadySet")
    fails(orc, prog(lambda p: p.add(M(1), M(2), M(3))), "UndefinedMemory")
    fails(orc, prog(lambda p: (p.add(K(0), K(0), M(1)), p.mul(M(2), M(1), M(0)))), "DivByZero")
    fails(orc, prog(lambda p: p.jump(M(0), K(3), FP(0))), "boolean")
    fails(orc, prog(lambda p: p.jump(K(1), K(100000), FP(0))), "PCOutOfBounds")
    fails(orc, prog(lambda p: p.poseidon16(FP(0), FP(8), FP(4))), "UndefinedMemory")
    fails(orc, prog(lambda p: (p.hint_witness("x", 1), p.add(M(1), K(0), M(2)))), "exhausted", {"x": []})
    fails(orc, prog(lambda p: (p.hint_witness("x", 1), p.add(M(1), K(0), M(2)))), "not all entries", {"x": [mont([1]), mont([2])]})
    fails(orc, prog(lambda p: (p.hint_debug_assert(M(0), K(5), "<"), p.add(M(0), K(0), K(5)))), "DebugAssert")
    fails(orc, prog(lambda p: p.extension_op("add", FP(0), FP(5), FP(10))), "InvalidExtensionOp|UndefinedMemory")


def test_bytecode_object(orc):
    p = Program()
    p.starting_frame_memory = 4
    p.add(K(0), K(5), M(0))
    p.return_from_main(1)
    bc = p.finalize()
    assert bc.log_size == 8 and bc.ending_pc == 255
    # Bytecode::hash = poseidon_compress_slice(instructions_multilinear, use_iv = true)
    h = np.zeros(8, dtype=np.uint32)
    for chunk in bc.multilinear.reshape(-1, 8):
        h = orc.poseidon16_compress(np.concatenate([h, chunk]))[0][:8]
    assert np.array_equal(bc.hash(), h)
    # undecodable rows are refused
    bad = bc.multilinear.copy()
    bad[0, 11] = mont([3])[0]          # precompile_data on an ADD
    with pytest.raises(lm.LmError, match="instruction 0"):
        vm.Bytecode(bad, bc.ending_pc, 4, [], {}).handle()


# ---- the aggregation program ----------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def program():
    return xa.build_program()


def test_xmss_program_counts_and_threads(orc, program):
    n = 6
    pi, w, info = xa.build_witness(program, n, np.random.default_rng(21))
    ex1 = execute(program, pi, w, n_threads=1)
    ex4 = execute(program, pi, w, n_threads=4)
    run = ob.VmRun(orc, program, pi, w)
    for ex in (ex1, ex4):
        assert np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps) and np.array_equal(ex.memory(), run.memory)
        assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.poseidon_calls(), ex1.poseidon_calls())
    L = program.info["layout"]
    n_tw = xa.TWEAK_TABLE_SIZE // 8 - 1
    assert ex1.n_poseidon_calls == 166 * n + n_tw + n + L["n_chunks"]       # per signature 2 + 110 + 22 + 32 (xmss.md)
    enc = info["sig"]["encoding"]
    n_copies = int((enc == 7).sum())                                        # chains that need no hash are copied (copy_5)
    assert ex1.n_extension_rows == n * (3 + 8) + n_copies + 2 + 2 + 1 + L["n_vars"]
    per_sig = (ex1.n_cycles - execute(program, *xa.build_witness(program, 2, np.random.default_rng(21))[:2]).n_cycles) / (n - 2)
    assert 480 <= per_sig <= 600                                            # 500-550 cycles per signature
    # the bytecode is looped: the loop body's rows are visited once per signature
    acc = np.bincount(ex1.pcs(), minlength=program.size)
    assert acc[program.labels["xmss_loop"]] == n + 1 and acc.max() == n + 1


def test_xmss_program_rejects_a_forged_signature(orc, program):
    pi, w, info = xa.build_witness(program, 3, np.random.default_rng(4))
    sig = {k: v.copy() for k, v in info["sig"].items()}
    sig["merkle_proof"][1, 17, 2] ^= 1
    _, w_bad, _ = xa.build_witness(program, 3, None, slot=info["slot"], sig=sig, message=info["message"])
    with pytest.raises(lm.LmError, match="ParallelSegmentFailed|MemoryAlreadySet"):
        execute(program, pi, w_bad)
    sig = {k: v.copy() for k, v in info["sig"].items()}
    sig["chain_tips"][0, 5, 1] ^= 1
    _, w_bad, _ = xa.build_witness(program, 3, None, slot=info["slot"], sig=sig, message=info["message"])
    with pytest.raises(lm.LmError, match="MemoryAlreadySet"):
        execute(program, pi, w_bad)


def test_xmss_program_trace_is_provable(orc, program):
    """the oracle's get_execution_trace of the run satisfies every AIR constraint, lookup and bus relation: the oracle proves it and
    both verifiers accept (a violated constraint fails the AIR final check, an inconsistent lookup the logup sum)"""
    pi, w, _ = xa.build_witness(program, 4, np.random.default_rng(8))
    ww = ob.VmRun(orc, program, pi, w).trace()
    assert ww["log_rows"] == {0: 12, 1: 8, 2: 10} or ww["log_rows"][0] >= 11
    b = ob.whir_builder(log_inv_rate=1, pow_bits=5, security=50)
    ob.set_threads(orc, 8)
    raw = ob.prove_execution(orc, ww, synth_witness.header(ww), b)
    ok, err = ob.verify_execution(orc, ww, raw, b)
    assert ok, err
    lb = lm.WhirBuilder.default(1, security_level=50, pow_bits=5)
    cfg = lm.WhirConfig.new(lb, synth_witness.stacked_n_vars(ww)).to_dict()
    sizes = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
    ok, err = lm.verify_execution(ww, lm.Prover.from_raw(raw, sizes), lb)
    assert ok, err
    assert int(from_monty(ww["bytecode_acc"])[:-1].max()) == 5              # bytecode_acc > 1: 4 iterations + the terminating call
    assert int(from_monty(ww["bytecode_acc"])[-1]) == (1 << ww["log_rows"][0]) - ww["non_padded"][0] + 1   # ending_pc: the padding rows


def test_runner_is_deterministic_under_repeated_parallel_runs(program):
    """The host pool runs parallel_for's with different participant counts back to back (memory resize: a few chunks; segments:
    all threads): 40 runs of the same input on 8 threads must all give the first run's log (a worker joining a generation it is
    not part of would run — and check out — twice)."""
    pi, w, _ = xa.build_witness(program, 40, np.random.default_rng(77))
    first = execute(program, pi, w, n_threads=8)
    ref = (first.pcs().tobytes(), first.fps().tobytes(), first.memory().tobytes(), first.poseidon_calls().tobytes(), first.extension_rows().tobytes())
    for k in range(40):
        ex = execute(program, pi, w, n_threads=8 if k % 3 else 5)
        got = (ex.pcs().tobytes(), ex.fps().tobytes(), ex.memory().tobytes(), ex.poseidon_calls().tobytes(), ex.extension_rows().tobytes())
        assert got == ref, k


def test_concurrent_runs_from_several_caller_threads(program):
    """Several provers of one process run the VM at the same time (bench.py --inflight, whole node): the pool's per-thread segment
    logs belong to ONE parallel batch at a time, so concurrent runs must take turns there and still give the sequential result."""
    import threading
    inputs = [xa.build_witness(program, 12 + 3 * t, np.random.default_rng(500 + t))[:2] for t in range(4)]
    def snapshot(ex):
        return (ex.pcs().tobytes(), ex.fps().tobytes(), ex.memory().tobytes(), ex.poseidon_calls().tobytes(), ex.extension_rows().tobytes())
    ref = [snapshot(execute(program, pi, w, n_threads=6)) for pi, w in inputs]
    bad = []
    def worker(t):
        pi, w = inputs[t]
        for k in range(12):
            if snapshot(execute(program, pi, w, n_threads=6)) != ref[t]:
                bad.append((t, k))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad
