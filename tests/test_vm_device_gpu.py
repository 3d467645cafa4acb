"""The leanVM's parallel loop batches on the device (lmh_execute_bytecode_device, csrc/lm_vm_device.hip) against the oracle's sequential
restatement (oracle/vm_oracle.hpp): the log a run leaves in HBM — pc / fp per cycle, the memory image with its defined mask, the
Poseidon call records, the ExtensionOp rows, the instruction counts — must equal the oracle's for every instruction kind, hint and
precompile variant executed INSIDE a segment, for deferred writes, for deref hints resolved after the run, for a sequential tail that
reads the segments' frames and for several batches in one program; a batch in which anything fails must give the host runner's error."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from leanmultisig_amd import vm
from leanmultisig_amd.programs import xmss_aggregate as xa
from leanmultisig_amd.vm import FP, K, M, Label, Program, Witness, execute, from_monty, to_monty
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

P = 0x7F000001
PI = to_monty(np.arange(1, 9)).astype(np.uint32)
FP0 = 10


def mont(x):
    return to_monty(x).astype(np.uint32)


def same_as_oracle(ctx, orc, bc, pi, w, expect_device=True, n_threads=2):
    ex = execute(bc, pi, w, n_threads=n_threads, ctx=ctx)
    assert ex.on_device == expect_device
    run = ob.VmRun(orc, bc, pi, w)
    assert ex.n_cycles == run.pcs.size and ex.memory_len == run.memory.size
    assert np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    bad = np.nonzero(ex.memory_defined() != run.defined)[0]
    assert bad.size == 0, f"defined mask differs at {bad[:8]}"
    bad = np.nonzero(ex.memory() != run.memory)[0]
    assert bad.size == 0, f"memory differs at {bad[:8]}: {ex.memory()[bad[:8]]} vs {run.memory[bad[:8]]}"
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows
    assert (ex.public_memory_size, ex.runtime_memory_size) == (run.public_memory_size, run.runtime_memory_size)
    # the same program on the host pool gives the same records (the oracle binding exposes only their count)
    ex_h = execute(bc, pi, w, n_threads=n_threads)
    assert not ex_h.on_device
    assert np.array_equal(ex.poseidon_calls(), ex_h.poseidon_calls()) and np.array_equal(ex.extension_rows(), ex_h.extension_rows())
    return ex, run


def loop_program(body, n_extra_args=0, frame_extra=0, after=None, second_loop=False, head=None, second_body=None, second_parallel=False):
    """main calls a PARALLEL loop over i in [0, n): frame = [ret, fp, i, end, out, perm, (extra args) | d, inv, nz, omnz, locals...];
    body(p, L) emits the iteration (L = first free frame offset); after(p) emits code behind the loop in main; second_loop: main then
    calls a second, sequential copy of the loop (one run_loop arms one batch, runner.rs:150-163: a later ParallelBatchStart is ignored);
    second_body / second_parallel: the second loop runs second_body over i in [0, n2) (hint "n2"), as a parallel batch of its own when
    second_parallel (handle_parallel_batch returns to a NEW run_loop, runner.rs:262-298, which arms the next ParallelBatchStart);
    head(p) emits code in main in front of the loop (main's frame grows to 200 cells, 64.. are free)."""
    p = Program()
    N, OUT, PERM, LF, LF2, N2 = 0, 1, 2, 3, 4, 5
    n_args = 4 + n_extra_args
    p.add(K(0), K(0), M(20))
    p.hint_witness("n", N)
    if second_body:
        p.hint_witness("n2", N2)
    p.hint_request_memory(OUT, M(N))
    p.hint_request_memory(PERM, M(N))
    p.hint_witness("perm", PERM, indirect=True)
    if head:
        head(p)

    def call(lf, label, ret, n_cell=N, frame="@frame"):
        p.hint_request_memory(lf, K(Label(frame)))
        p.deref(lf, 0, K(Label(ret)))
        p.deref(lf, 1, FP(0))
        p.deref(lf, 2, K(0))
        p.deref(lf, 3, M(n_cell))
        p.deref(lf, 4, M(OUT))
        p.deref(lf, 5, M(PERM))
        for k in range(n_extra_args):
            p.deref(lf, 6 + k, K(1000 + k))
        p.jump(K(1), K(Label(label)), M(lf))
        p.label(ret)

    call(LF, "loop", "after")
    if second_loop:
        call(LF2, "loop_b", "after2", N2 if second_body else N, "@frame_b" if second_body else "@frame")
    if after:
        after(p)
    p.return_from_main(21)
    p.starting_frame_memory = 200 if head else 64
    I, END = 2, 3
    d, inv, nz, omnz = 2 + n_args, 3 + n_args, 4 + n_args, 5 + n_args
    L = 6 + n_args

    def emit_loop(tag, parallel, body=body, frame_label="@frame"):
        if parallel:
            p.hint_parallel_batch_start(n_args, M(END))
        p.label("loop" + tag)
        p.add(M(d), M(END), M(I))
        p.hint_inverse(M(d), inv)
        p.mul(M(d), M(inv), M(nz))
        p.add(M(omnz), M(nz), K(1))
        p.mul(M(omnz), M(d), K(0))
        p.jump(M(nz), K(Label("body" + tag)), FP(0))
        p.jump(K(1), M(0), M(1))
        p.label("body" + tag)
        used = body(p, L)
        nxt = L + used
        ip1 = nxt + 1
        frame = ip1 + 1 + frame_extra
        p.hint_request_memory(nxt, K(frame))
        p.deref(nxt, 0, M(0))
        p.deref(nxt, 1, M(1))
        p.add(M(I), K(1), M(ip1))
        p.deref(nxt, 2, M(ip1))
        for a in range(3, 2 + n_args):
            p.deref(nxt, a, M(a))
        p.jump(K(1), K(Label("loop" + tag)), M(nxt))
        p.labels[frame_label] = frame

    emit_loop("", True)
    if second_loop:
        emit_loop("_b", second_parallel, second_body or body, "@frame_b" if second_body else "@frame")
    return p


def hash_and_store_body(p, L):
    """two chained hashes of a hinted block, i stored into out[perm[i]] (a deferred write)"""
    blk, h, t, idx, o = L, L + 8, L + 24, L + 25, L + 26
    p.hint_witness("block", blk)
    p.poseidon16(FP(blk), FP(blk), FP(h))
    p.poseidon16(FP(h), FP(blk), FP(h + 8))
    p.add(M(5), M(2), M(t))
    p.deref(t, 0, M(idx))
    p.add(M(4), M(idx), M(o))
    p.deref(o, 0, M(2))
    return 27


def hints_for(n, rng, extra=None):
    h = {"n": [mont([n])], "perm": [mont(rng.permutation(n))]}
    if extra is None:
        h["block"] = [ob.rand_field(rng, 8) for _ in range(n)]
    h.update(extra or {})
    return h


@pytest.mark.parametrize("n", [33, 200, 1000])
def test_batch_on_the_device_equals_oracle(ctx, orc, n):
    rng = np.random.default_rng(n)
    bc = loop_program(hash_and_store_body).finalize()
    w = Witness(bc, 0, hints_for(n, rng))
    ex, run = same_as_oracle(ctx, orc, bc, PI, w)
    assert ex.n_poseidon_calls == 2 * n


def test_small_batch_stays_on_the_host(ctx, orc):
    bc = loop_program(hash_and_store_body).finalize()
    w = Witness(bc, 0, hints_for(9, np.random.default_rng(1)))
    same_as_oracle(ctx, orc, bc, PI, w, expect_device=False)


def kitchen_sink_body(p, L):
    """every instruction kind with its unknown-operand cases, every hint and every precompile variant inside a segment"""
    c = L
    # ADD / MUL with each operand unknown, constants, fp-relative operands
    p.add(K(3), M(2), M(c))                  # c0 = 3 + i
    p.add(M(c + 1), K(5), M(c))              # a unknown: c1 = c0 - 5
    p.add(K(7), M(c + 2), M(c))              # c unknown: c2 = c0 - 7
    p.mul(M(c), K(6), M(c + 3))              # c3 = 6 c0
    p.mul(M(c + 4), M(c), M(c + 3))          # a unknown: c4 = c3 / c0 = 6   (c0 = 3 + i != 0)
    p.mul(M(c), M(c + 5), M(c + 3))          # c unknown: c5 = 6
    p.add(K(0), FP(c + 40), M(c + 6))        # pointer into the own frame
    p.mul(K(2), FP(0), M(c + 7))
    # DEREF: load from the shared prefix, store into the own frame, store fp-relative
    p.add(K(0), K(3), M(c + 8))
    p.deref(c + 8, 2, M(c + 9))              # m[5] = 6 (public input)
    p.deref(c + 6, 1, K(99))                 # m[fp + c + 41] = 99
    p.deref(c + 6, 2, M(c + 9))
    p.deref(c + 6, 3, FP(5))
    p.range_check(c + 9, K(9), c + 10)       # 3 cells: c+10 .. c+12 (deref hints, resolved after the run)
    p.add(K(0), K(1 << 20), M(c + 13))       # an address nobody writes: zero-filled by resolve_deref_hints
    p.hint_deref(c + 13, c + 14)
    p.deref(c + 13, 0, M(c + 14))
    # hints
    p.hint_inverse(K(0), c + 15)
    p.add(M(c + 15), K(0), K(0))
    p.add(K(0), K(0b1011_0110_1100_0011_1010_0101 + (5 << 24)), M(c + 16))
    p.add(K(0), K(77), M(c + 17))
    p.hint_decompose_bits_xmss(FP(c + 50), FP(c + 16), K(2), K(6))          # 8 cells
    p.hint_decompose_bits_merkle_whir(FP(c + 58), M(c + 16), K(8))          # 3 cells
    p.hint_decompose_bits(M(c + 17), FP(c + 62), K(8))                      # 8 cells
    p.hint_less_than(M(c + 17), K(78), M(c + 18))
    p.hint_less_than(M(2), M(3), M(c + 19))
    p.hint_log2_ceil(M(c + 17), M(c + 20))
    p.hint_debug_assert(M(c + 17), K(78), "<")
    p.hint_debug_assert(M(c + 17), K(100), "<=", preceds_runtime_inequality=True)
    p.hint_witness("ef", c + 70)                                             # a, b: 10 cells
    p.add(K(0), FP(c + 80), M(c + 21))
    p.hint_witness("vec", c + 21, indirect=True)                             # 15 cells at c+80
    p.add(M(c + 70), M(c + 80), M(c + 22))
    # jump not taken / taken inside the iteration
    p.jump(K(0), K(Label("never")), FP(0))
    p.jump(M(c + 18), K(Label("ks_on")), FP(0))
    p.label("never")
    p.panic()
    p.label("ks_on")
    # ExtensionOp: every mode, lengths > 1, unknown solving, copy_5 (ONE in the extension field from the prefix is not available: build it)
    A, B = c + 70, c + 75
    for k, v in enumerate([1, 0, 0, 0, 0]):
        p.add(K(0), K(v), M(c + 100 + k))
    p.extension_op("mul", FP(A), FP(B), FP(c + 105))
    p.extension_op("mul", FP(c + 110), FP(B), FP(c + 105))                   # A unknown
    p.extension_op("mul", FP(A), FP(c + 115), FP(c + 105))                   # B unknown
    p.extension_op("add", FP(A), FP(B), FP(c + 120))
    p.extension_op("add", FP(c + 125), FP(B), FP(c + 120))                   # A unknown
    p.extension_op("mul", FP(A), FP(c + 100), FP(c + 130))                   # copy_5
    p.extension_op("mul", FP(c + 135), FP(c + 100), FP(c + 140))             # both unknown: zeros
    p.extension_op("mul", FP(c + 80), FP(c + 80), FP(c + 145), size=3)       # dot product of 3 pairs
    p.extension_op("poly_eq", FP(c + 80), FP(c + 85), FP(c + 150), size=2)
    p.extension_op("mul", FP(c + 50), FP(c + 80), FP(c + 155), size=3, is_be=True)
    p.extension_op("add", FP(c + 50), FP(c + 80), FP(c + 160), size=2, is_be=True)
    p.extension_op("poly_eq", FP(c + 62), FP(c + 80), FP(c + 165), size=2, is_be=True)
    # Poseidon variants: compress, half output, hardcoded left (from the public input), permute
    p.poseidon16(FP(c + 80), FP(c + 87), FP(c + 170))
    p.add(K(0), FP(c + 80), M(c + 23))
    p.add(K(0), FP(c + 170), M(c + 24))
    p.poseidon16(M(c + 23), M(c + 24), FP(c + 180), half=True)
    p.poseidon16(M(c + 23), K(0), FP(c + 188), left=4)
    p.poseidon16(FP(c + 170), FP(c + 80), FP(c + 196), half=True, left=2)
    p.poseidon16(FP(c + 80), FP(c + 170), FP(c + 204), permute=True)
    # a deferred write of a hash into the shared output array
    p.add(M(5), M(2), M(c + 25))
    p.deref(c + 25, 0, M(c + 26))
    p.add(M(4), M(c + 26), M(c + 27))
    p.deref(c + 27, 0, M(c + 204))
    return 224


@pytest.mark.parametrize("n", [40, 300])
def test_every_instruction_hint_and_precompile_inside_a_segment(ctx, orc, n):
    rng = np.random.default_rng(100 + n)
    bc = loop_program(kitchen_sink_body, n_extra_args=3).finalize()
    extra = {"ef": [ob.rand_field(rng, 10) for _ in range(n)], "vec": [ob.rand_field(rng, 15) for _ in range(n)]}
    w = Witness(bc, 0, hints_for(n, rng, extra))
    ex, run = same_as_oracle(ctx, orc, bc, PI, w)
    assert ex.n_extension_rows == n * (7 + 3 + 2 + 3 + 2 + 2) and ex.n_poseidon_calls == 5 * n


def test_sequential_tail_reads_the_segment_frames(ctx, orc):
    """main, after the loop, loads a cell of a segment's frame: the frames live in the device image only until then"""
    n = 64

    def after(p):
        p.add(M(3), K(Label("@frame")), M(30))     # loop frame of iteration 1 = first segment
        p.deref(30, 14, M(31))                     # its first hash word (blk = L = 10 -> h = 18; + 0): frame offset 18 ... any defined cell
        p.add(M(31), K(1), M(32))

    rng = np.random.default_rng(3)
    bc = loop_program(hash_and_store_body, after=after).finalize()
    w = Witness(bc, 0, hints_for(n, rng))
    ex, run = same_as_oracle(ctx, orc, bc, PI, w)
    m = from_monty(ex.memory())
    assert m[FP0 + 32] == (m[FP0 + 31] + 1) % P and ex.memory_defined()[FP0 + 31]


def test_sequential_loop_behind_the_device_batch(ctx, orc):
    """after the batch main runs a second, sequential loop: host cycles, Poseidon calls and deferred derefs behind the device's in the
    log, hint cursors continued where the segments left them"""
    n = 48
    rng = np.random.default_rng(4)
    bc = loop_program(hash_and_store_body, second_loop=True).finalize()
    h = hints_for(n, rng)
    h["block"] = h["block"] + h["block"]           # the second loop hashes the same blocks again and stores the same values
    w = Witness(bc, 0, h)
    ex, run = same_as_oracle(ctx, orc, bc, PI, w)
    assert ex.n_poseidon_calls == 4 * n


def frame_reader_body(p, L):
    """iteration i of the SECOND loop reads a hash word of the first loop's segment i + 1 (through main's frame pointer, saved in the
    call frame: main's cell 3 holds the first loop's frame of iteration 0) and stores a function of it"""
    lf, off, seg, v = L, L + 1, L + 2, L + 3
    p.deref(1, 3, M(lf))                                   # m[main fp + 3] = frame of iteration 0 of the first loop
    p.add(M(2), K(1), M(off))
    p.mul(M(off), K(Label("@frame")), M(off + 10))         # (i + 1) * frame size
    p.add(M(lf), M(off + 10), M(seg))
    p.deref(seg, 18, M(v))                                 # first hash word of that segment (hash_and_store_body: h = L + 8 = 18)
    p.add(M(v), K(7), M(v + 1))
    p.poseidon16(M(seg), M(seg), FP(L + 16))               # and hashes the start of the segment's frame
    return 24


@pytest.mark.parametrize("n2,second_parallel", [(9, True), (9, False), (40, True)])
def test_host_batch_behind_a_device_batch_reads_its_frames(ctx, orc, n2, second_parallel):
    """Round-4 advisor finding: a batch that does not qualify for the device (fewer than 32 segments) behind one that ran there.  Its
    segments read the frames of the first batch, which exist only in the device image until the window is closed: the host batch must
    see them (csrc/host/lm_vm.cpp: execute_impl closes the windows before every host batch).  n2 = 40: both batches on the device, the
    second one reads the first one's frames from the image."""
    n = 64
    rng = np.random.default_rng(5)
    bc = loop_program(hash_and_store_body, second_loop=True, second_body=frame_reader_body, second_parallel=second_parallel).finalize()
    h = hints_for(n, rng)
    h["n2"] = [mont([n2])]
    ex, run = same_as_oracle(ctx, orc, bc, PI, Witness(bc, 0, h))
    assert ex.n_poseidon_calls == 2 * n + n2


def chain_head(n_links, expected, read_in_body=False):
    """main hashes a chain of n_links compressions over a hinted block in front of the loop and checks the last digest against five
    expected words (copy_5): with a device context the runner records these calls and the check (csrc/host/lm_vm.cpp: MemBuf) and
    executes them while the segments run"""
    def head(p):
        p.hint_witness("head_block", 64)                       # 16 words at 64..80
        p.poseidon16(FP(64), FP(72), FP(80))
        for k in range(1, n_links):
            p.poseidon16(FP(80 + 8 * (k - 1)), FP(72), FP(80 + 8 * k))
        last = 80 + 8 * (n_links - 1)
        p.add(K(0), K(1), M(190))
        for k in range(4):
            p.add(K(0), K(0), M(191 + k))
        for k in range(5):
            p.add(K(0), K(int(expected[k])), M(195 + k))
        p.extension_op("mul", FP(last), FP(190), FP(195))
    return head


def chain_digest(orc, block, n_links):
    h = orc.poseidon16_compress(np.concatenate([block[:8], block[8:]]))[0][:8]
    for _ in range(1, n_links):
        h = orc.poseidon16_compress(np.concatenate([h, block[8:]]))[0][:8]
    return from_monty(h)


def test_hash_chain_and_its_check_in_front_of_the_device_batch(ctx, orc):
    n, links = 100, 12
    rng = np.random.default_rng(17)
    block = ob.rand_field(rng, 16)
    d = chain_digest(orc, block, links)
    hints = hints_for(n, rng)
    hints["head_block"] = [block]
    bc = loop_program(hash_and_store_body, head=chain_head(links, d)).finalize()
    ex, run = same_as_oracle(ctx, orc, bc, PI, Witness(bc, 0, hints))
    assert ex.n_poseidon_calls == 2 * n + links and ex.n_extension_rows == 1
    # the expected digest is wrong: the check fails while the segments run, and the error is the host runner's
    bad = d.copy()
    bad[2] = (bad[2] + 1) % P
    bc = loop_program(hash_and_store_body, head=chain_head(links, bad)).finalize()
    w = Witness(bc, 0, hints)
    with pytest.raises(lm.LmError, match="NotEqual") as dev:
        execute(bc, PI, w, n_threads=2, ctx=ctx)
    with pytest.raises(lm.LmError) as host:
        execute(bc, PI, w, n_threads=2)
    assert str(dev.value) == str(host.value)


@pytest.mark.parametrize("use", ["strict", "deref_only", "solve"])
def test_segments_reading_a_pending_digest_send_the_batch_to_the_host(ctx, orc, use):
    """(use = deref_only / solve: round-4 advisor finding — reads that TOLERATE None: a DEREF whose result nobody reads strictly, and an
    ADD that would solve for the cell the DEREF left undefined; the image carries VM_PENDING in such cells, not None.)
    iteration i reads a word of the chain's digest number which[i] through main's frame pointer (saved in its call frame).  Iteration
    0 — run by the host, which executes what it needs of the chain — reads link 0; the later links are still None in the image the
    segments get: they fail there, and the batch runs on the host pool — same log as the oracle's"""
    n, links = 64, 5
    rng = np.random.default_rng(18)
    block = ob.rand_field(rng, 16)
    d = chain_digest(orc, block, links)

    def body(p, L):
        used = hash_and_store_body(p, L)
        w, w8, ptr, v = L + used, L + used + 1, L + used + 2, L + used + 3
        p.hint_witness("which", w)
        p.mul(M(w), K(8), M(w8))
        p.add(M(1), M(w8), M(ptr))                            # main's fp + 8 * which
        p.deref(ptr, 80 + 3, M(v))                            # word 3 of that link's digest
        if use == "strict":
            p.add(M(v), K(1), M(v + 1))
        elif use == "solve":                                  # with v None the ADD would DEFINE v = expect - 1 instead of CHECKING v + 1 == expect
            p.hint_witness("expect", v + 1)
            p.add(M(v), K(1), M(v + 1))
        return used + 6

    hints = hints_for(n, rng)
    hints["head_block"] = [block]
    hints["which"] = [mont([0])] + [mont([int(x)]) for x in rng.integers(1, links, n - 1)]
    bc = loop_program(body, head=chain_head(links, d)).finalize()
    if use == "solve":
        # link k's digest word 3, + 1 — wrong in one iteration: the reference's check fails there, and so must this run (same error text)
        digests = [chain_digest(orc, block, k + 1) for k in range(links)]
        which = [int(from_monty(x)[0]) for x in hints["which"]]
        hints["expect"] = [mont([(int(digests[k][3]) + 1) % P]) for k in which]
        same_as_oracle(ctx, orc, bc, PI, Witness(bc, 0, hints), expect_device=False)
        hints["expect"][n // 2] = mont([(int(digests[which[n // 2]][3]) + 2) % P])
        w = Witness(bc, 0, hints)
        with pytest.raises(lm.LmError, match="ParallelSegmentFailed") as dev:
            execute(bc, PI, w, n_threads=2, ctx=ctx)
        with pytest.raises(lm.LmError) as host:
            execute(bc, PI, w, n_threads=2)
        assert str(dev.value) == str(host.value)
        return
    same_as_oracle(ctx, orc, bc, PI, Witness(bc, 0, hints), expect_device=False)


def test_conflicting_deferred_writes_give_the_host_runners_error(ctx, orc):
    n = 40
    rng = np.random.default_rng(0)
    perm = np.arange(n)
    perm[-1] = perm[-2]                              # two iterations store different values into one cell
    bc = loop_program(hash_and_store_body).finalize()
    hints = {"n": [mont([n])], "perm": [mont(perm)], "block": [ob.rand_field(rng, 8) for _ in range(n)]}
    w = Witness(bc, 0, hints)
    with pytest.raises(lm.LmError, match="MemoryAlreadySet") as dev:
        execute(bc, PI, w, n_threads=2, ctx=ctx)
    with pytest.raises(lm.LmError, match="MemoryAlreadySet") as host:
        execute(bc, PI, w, n_threads=2)
    assert str(dev.value) == str(host.value)


def test_failing_segment_gives_the_host_runners_error(ctx, orc):
    def body(p, L):
        used = hash_and_store_body(p, L)
        p.add(K(0), K(35), M(L + used))
        p.hint_debug_assert(M(2), M(L + used), "<")   # i < 35 fails in the later segments
        p.add(M(L + used + 1), M(2), K(36))
        p.mul(M(L + used + 2), M(L + used + 1), K(1))  # division by 36 - i: DivByZero... only at i = 36
        return used + 3

    n = 50
    bc = loop_program(body).finalize()
    w = Witness(bc, 0, hints_for(n, np.random.default_rng(8)))
    with pytest.raises(lm.LmError) as dev:
        execute(bc, PI, w, n_threads=2, ctx=ctx)
    with pytest.raises(lm.LmError) as host:
        execute(bc, PI, w, n_threads=2)
    assert str(dev.value) == str(host.value) and "ParallelSegmentFailed" in str(dev.value)


@pytest.fixture(scope="module")
def program():
    return xa.build_program()


def test_xmss_program_on_the_device(ctx, orc, program):
    pi, w, _ = xa.build_witness(program, 70, np.random.default_rng(21), slot=0x0BADCAFE)
    ex, run = same_as_oracle(ctx, orc, program, pi, w)
    assert ex.n_poseidon_calls == run.n_poseidon_calls
    # the tables lmh_get_execution_trace builds from the resident log equal the oracle's get_execution_trace
    ex2 = execute(program, pi, w, ctx=ctx, lazy=True)
    assert ex2.on_device
    ref = run.trace()
    dt = vm.DeviceTrace(ctx, program, ex2, pi)
    assert dt.view.log_memory == ref["log_memory"]
    assert np.array_equal(dt.memory(), ref["memory"])
    for t in range(3):
        assert dt.view.tables[t].log_rows == ref["log_rows"][t] and dt.view.tables[t].non_padded_n_rows == ref["non_padded"][t]
        got, want = dt.table(t), ref["tables"][t]
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, f"table {t}: columns {bad[:10]} differ"
    dt.close()
    ex2.close()


def test_xmss_forged_signature_is_rejected_with_the_host_runners_error(ctx, program):
    pi, w, info = xa.build_witness(program, 40, np.random.default_rng(22))
    for field, at in (("merkle_proof", (17, 9, 2)), ("chain_tips", (31, 5, 1))):
        sig = {k: v.copy() for k, v in info["sig"].items()}
        sig[field][at] ^= 1
        _, w_bad, _ = xa.build_witness(program, 40, None, slot=info["slot"], sig=sig, message=info["message"])
        with pytest.raises(lm.LmError, match="ParallelSegmentFailed|MemoryAlreadySet") as dev:
            execute(program, pi, w_bad, n_threads=2, ctx=ctx)
        with pytest.raises(lm.LmError) as host:
            execute(program, pi, w_bad, n_threads=2)
        assert str(dev.value) == str(host.value)
