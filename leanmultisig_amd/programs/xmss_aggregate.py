"""The XMSS aggregation program (type-1, raw signatures only) assembled by hand at the ISA level, and the witness of one run.

The reference compiles this program from its zkDSL (crates/rec_aggregation/zkdsl_implem/{main,xmss_aggregate,hashing,utils}.py);
the compiler is out of scope here, so the same program is written directly in leanVM instructions with
leanmultisig_amd/vm.py.  What is kept from the reference, statement by statement:
  main.py:38-57,119-175,235-245   main(): preamble memory, the hinted public-input buffer, type-1 checks, tweak table + its hash,
                                  pubkeys + their hash (slice_hash_with_iv_dynamic_unroll: binary decomposition of the count,
                                  one unrolled block per bit), the PARALLEL loop over the raw signatures, the zero bytecode claim,
                                  the hash of the input buffer onto the public input
  xmss_aggregate.py:42-116        xmss_verify: encode (2 hashes), hinted 6-bit decomposition with range checks and the
                                  "remaining < 127" uniformity check, 21 chain pairs dispatched through 64-entry jump tables
                                  (match_range), target sum, wots_pk_hash (22 hashes), xmss_merkle_verify (8 chunks of 4 levels
                                  dispatched through 16-entry jump tables, hinted siblings, the last hash written onto the public key)
  the memory layout of hashing.py:7-17 / compilation.rs:20-25 (zero vector, sampling domain separator, one, repeated ones, tweak
  table at PREAMBLE_MEMORY_END), the public-input layout of TYPE1_TYPE2_LAYOUT.md, the hint streams of type_1_aggregation.rs:317-356.
What is NOT here: the recursion branches (children proofs, type-2, split) — their code is the in-VM WHIR verifier (whir.py,
recursion.py); a run with n_recursions != 0 fails on an assertion.  The reference's bytecode is 2^19 instructions because of
that verifier; `build_program(log_size=19)` pads this one to the same table size with unreachable instructions, as the compiler
pads its own to a power of two (c_compile_final.rs:102-110).
Lowering conventions are this file's own (the reference's register allocation is not reproduced); every instruction is one of
the reference's four kinds and the cycle count per signature comes out at ~520 (the survey's estimate for the compiled DSL: 500-550).
"""
import numpy as np

from ..vm import FP, K, M, Label, Program, Witness, to_monty
from ..xmss import (CHAIN_LENGTH, LOG_LIFETIME, TARGET_SUM, TWEAK_CHAIN, TWEAK_ENCODING, TWEAK_MERKLE, TWEAK_WOTS_PK, V, Xmss,
                    from_monty, make_tweak, rand_field)

P = 0x7F000001
DIM, DIGEST_LEN = 5, 8
# ---- memory layout (hashing.py:7-17, compilation.rs:20-25) ---------------------------------------------------------------------
PUBLIC_INPUT_LEN = DIGEST_LEN
ZERO_VEC_PTR, ZERO_VEC_LEN = PUBLIC_INPUT_LEN, 16
SDS_PTR = ZERO_VEC_PTR + ZERO_VEC_LEN
ONE_EF_PTR = SDS_PTR + DIGEST_LEN
REPEATED_ONES_PTR, NUM_REPEATED_ONES = ONE_EF_PTR + DIM, 32
TWEAK_TABLE_ADDR = REPEATED_ONES_PTR + NUM_REPEATED_ONES
N_TWEAKS = 1 + V * CHAIN_LENGTH + 1 + LOG_LIFETIME
TWEAK_LEN = 4
TWEAK_TABLE_SIZE = (N_TWEAKS * TWEAK_LEN + 7) // 8 * 8
PREAMBLE_MEMORY_LEN = ZERO_VEC_LEN + DIGEST_LEN + DIM + NUM_REPEATED_ONES + TWEAK_TABLE_SIZE
MAIN_FP = (PUBLIC_INPUT_LEN + PREAMBLE_MEMORY_LEN + 4) // 5 * 5        # the runner's starting fp (runner.rs:253-255)
TW_ENC = TWEAK_TABLE_ADDR
TW_CHAIN = TW_ENC + TWEAK_LEN
TW_WOTS_PK = TW_CHAIN + V * CHAIN_LENGTH * TWEAK_LEN
TW_MERKLE = TW_WOTS_PK + TWEAK_LEN
# ---- xmss (xmss_aggregate.py:4-32) ----------------------------------------------------------------------------------------------
XMSS_DIGEST_LEN, RANDOMNESS_LEN, PUB_KEY_SIZE = 4, 6, 8
WOTS_SIG_SIZE = RANDOMNESS_LEN + V * XMSS_DIGEST_LEN
WOTS_PK_PAIR_STRIDE = 10
NUM_ENCODING_FE = 6            # div_ceil(V, 24 / W)
N_MERKLE_CHUNKS = LOG_LIFETIME // 4
MAX_N_SIGS = 1 << 15           # MAX_XMSS_AGGREGATED (compilation.rs:43), = MAX_N_DUPS
LOG_MAX_N_SIGS = 15
TYPE_1_FLAG = 1


def layout(log_size):
    """offsets inside the public-input buffer (main.py:14-31, TYPE1_TYPE2_LAYOUT.md) for a bytecode of 2^log_size instructions"""
    n_vars = log_size + 4
    claim_size = (n_vars + 1) * DIM
    claim_padded = (claim_size + 7) // 8 * 8
    domsep = DIGEST_LEN + claim_padded
    comp = domsep + DIGEST_LEN
    return dict(n_vars=n_vars, claim_size=claim_size, claim_padded=claim_padded, claim=DIGEST_LEN, domsep=domsep, pubkeys_hash=comp,
                message=comp + 8, merkle_chunks=comp + 16, tweaks_hash=comp + 24, size=comp + 32, n_chunks=(comp + 32) // 8)


class Frame:
    """bump allocator of fp-relative cells"""

    def __init__(self, start=0):
        self.top = start

    def alloc(self, n=1):
        o = self.top
        self.top += n
        return o


def build_program(log_size=None):
    """-> vm.Bytecode.  log_size = None: the smallest power of two; the layout constants depend on it (self-referential in the
    reference: compile_main_program_self_referential, compilation.rs:86-105), so the program is assembled for a given size."""
    if log_size is None:
        for guess in range(17, 22):
            bc = _assemble(guess, pad=False)
            if bc.log_size == guess:
                return bc
        raise RuntimeError("no fixed point for the bytecode size")
    return _assemble(log_size, pad=True)


def _assemble(log_size, pad):
    L = layout(log_size)
    p = Program()
    f = Frame()
    A = lambda off: MAIN_FP + off  # noqa: E731 — absolute address of a main-frame cell (main's fp is fixed by the memory layout)

    def const_cell(value):
        c = f.alloc()
        p.add(K(0), K(value), M(c))
        return c

    def ptr_plus(cell, k):
        c = f.alloc()
        p.add(M(cell), K(k), M(c))
        return c

    def copy_5(src, dst):  # dot_product_ee(src, ONE_EF_PTR, dst) (utils.py:359-361); src: M / K address, dst: any address operand
        p.extension_op("mul", src, K(ONE_EF_PTR), dst)

    # ================================================================ main ================================================================
    # build_preamble_memory (utils.py:11-29).  The first instruction's operand_a is 0 (BYTECODE_ZERO_EVAL, compilation.rs:88).
    zv = const_cell(ZERO_VEC_PTR)
    for i in range(ZERO_VEC_LEN):
        p.deref(zv, i, K(0))
    sds = const_cell(SDS_PTR)
    p.deref(sds, 0, K(1))
    for i in range(1, DIGEST_LEN):
        p.deref(sds, i, K(0))
    one = const_cell(ONE_EF_PTR)
    p.deref(one, 0, K(1))
    for i in range(1, DIM):
        p.deref(one, i, K(0))
    ones = const_cell(REPEATED_ONES_PTR)
    for i in range(NUM_REPEATED_ONES):
        p.deref(ones, i, K(1))
    # input buffer
    n_chunks = f.alloc()
    p.hint_witness("input_data_num_chunks", n_chunks)
    size8 = f.alloc()
    p.mul(M(n_chunks), K(DIGEST_LEN), M(size8))
    data = f.alloc()
    p.hint_request_memory(data, M(size8))
    p.hint_witness("input_data", data, indirect=True)
    d2 = ptr_plus(data, 2)
    copy_5(M(d2), K(ZERO_VEC_PTR))                 # set_to_6_zeros(data_buf + 2)
    p.deref(data, 7, K(0))
    disc = f.alloc()
    p.deref(data, 0, M(disc))
    not_disc = f.alloc()
    p.add(M(not_disc), M(disc), K(1))              # discriminator == TYPE_2_FLAG (0) -> the type-2 branch (not assembled)
    p.jump(M(not_disc), K(Label("unsupported")), FP(0))
    p.add(M(disc), K(0), K(TYPE_1_FLAG))           # assert discriminator == TYPE_1_FLAG
    is_split = f.alloc()
    p.hint_witness("is_split", is_split)
    p.jump(M(is_split), K(Label("unsupported")), FP(0))
    n_sigs = f.alloc()
    p.deref(data, 1, M(n_sigs))
    inv = f.alloc()
    p.hint_inverse(M(n_sigs), inv)
    p.mul(M(n_sigs), M(inv), K(1))                 # assert n_sigs != 0
    nm1 = f.alloc()
    p.add(M(nm1), K(1), M(n_sigs))
    p.range_check(nm1, K(MAX_N_SIGS - 1), f.alloc(3))   # assert n_sigs - 1 < MAX_N_SIGS
    tw = const_cell(TWEAK_TABLE_ADDR)
    p.hint_witness("tweak_table", tw, indirect=True)
    pk_hash_exp, msg_ptr, mchunks_ptr, tw_hash_exp = (ptr_plus(data, L[k]) for k in ("pubkeys_hash", "message", "merkle_chunks", "tweaks_hash"))
    meta = f.alloc(3)
    p.hint_witness("meta", meta)
    p.add(M(meta), K(0), K(0))                     # n_recursions == 0: the recursion branches are not assembled
    p.range_check(meta + 1, K(MAX_N_SIGS - 1), f.alloc(3))   # n_dup < MAX_N_DUPS
    n_total = f.alloc()
    p.add(M(n_sigs), M(meta + 1), M(n_total))
    pk_words = f.alloc()
    p.mul(M(n_total), K(PUB_KEY_SIZE), M(pk_words))
    all_pk = f.alloc()
    p.hint_request_memory(all_pk, M(pk_words))
    p.hint_witness("pubkeys", all_pk, indirect=True)
    raw_idx = f.alloc()
    p.hint_request_memory(raw_idx, M(meta + 2))
    p.hint_witness("raw_indices", raw_idx, indirect=True)
    agg_sizes = f.alloc()
    p.hint_request_memory(agg_sizes, M(meta))
    p.hint_witness("aggregate_sizes", agg_sizes, indirect=True)
    p.add(M(agg_sizes), K(0), M(f.alloc()))        # (anchors the hints above on an instruction)
    # computed_tweaks_hash = slice_hash(tweak_table, chunks) (hashing.py:69-75); copy_8 onto the expected hash
    n_tw_chunks = TWEAK_TABLE_SIZE // DIGEST_LEN
    tw_states = f.alloc((n_tw_chunks - 1) * DIGEST_LEN)
    p.poseidon16(K(TWEAK_TABLE_ADDR), K(TWEAK_TABLE_ADDR + 8), K(A(tw_states)))
    for j in range(1, n_tw_chunks - 1):
        p.poseidon16(K(A(tw_states + 8 * (j - 1))), K(TWEAK_TABLE_ADDR + 8 * (j + 1)), K(A(tw_states + 8 * j)))
    tw_hash = A(tw_states + 8 * (n_tw_chunks - 2))
    copy_5(K(tw_hash), M(tw_hash_exp))
    copy_5(K(tw_hash + 3), M(ptr_plus(tw_hash_exp, 3)))
    # computed_pubkeys_hash = slice_hash_with_iv_dynamic_unroll(all_pubkeys, n_sigs, 15) (hashing.py:99-118)
    states = f.alloc()
    p.hint_request_memory(states, M(pk_words))
    p.poseidon16(K(ZERO_VEC_PTR), M(all_pk), M(states))
    bits = f.alloc(LOG_MAX_N_SIGS)                 # big-endian bits of n_iters = n_sigs - 1
    p.hint_decompose_bits(M(nm1), FP(bits), K(LOG_MAX_N_SIGS))
    acc = None
    for k in range(LOG_MAX_N_SIGS):                # bits[k] has weight 2^(14 - k)
        p.mul(M(bits + k), M(bits + k), M(bits + k))
        t = f.alloc()
        p.mul(M(bits + k), K(1 << (LOG_MAX_N_SIGS - 1 - k)), M(t))
        if acc is None:
            acc = t
        else:
            s = f.alloc()
            p.add(M(acc), M(t), M(s))
            acc = s
    p.add(M(acc), K(0), M(nm1))                    # the decomposition is n_sigs - 1
    cur_s = states
    cur_d = ptr_plus(all_pk, DIGEST_LEN)
    for k in range(LOG_MAX_N_SIGS):
        n_it = 1 << (LOG_MAX_N_SIGS - 1 - k)
        nxt_s, nxt_d = f.alloc(), f.alloc()
        p.jump(M(bits + k), K(Label(f"pkh_do_{k}")), FP(0))
        p.add(M(cur_s), K(0), M(nxt_s))
        p.add(M(cur_d), K(0), M(nxt_d))
        p.jump(K(1), K(Label(f"pkh_after_{k}")), FP(0))
        p.label(f"pkh_do_{k}")
        s_prev, d_prev = cur_s, cur_d              # block of n_it unrolled iterations: new_state = state_ptr + DIGEST_LEN each
        for t in range(n_it):
            s_new = nxt_s if t == n_it - 1 else f.alloc()
            p.add(M(s_prev), K(DIGEST_LEN), M(s_new))
            p.poseidon16(M(s_prev), M(d_prev), M(s_new))
            d_new = nxt_d if t == n_it - 1 else f.alloc()
            p.add(M(d_prev), K(DIGEST_LEN), M(d_new))
            s_prev, d_prev = s_new, d_new
        p.label(f"pkh_after_{k}")
        cur_s, cur_d = nxt_s, nxt_d
    copy_5(M(cur_s), M(pk_hash_exp))               # copy_8(computed_pubkeys_hash, pubkeys_hash_expected)
    copy_5(M(ptr_plus(cur_s, 3)), M(ptr_plus(pk_hash_exp, 3)))
    buffer = f.alloc()
    p.hint_request_memory(buffer, M(n_total))
    ntm1 = f.alloc()
    p.add(M(ntm1), K(1), M(n_total))
    # ---- for i in parallel_range(0, n_raw_xmss) (main.py:161-167): call the loop function -------------------------------------------
    LOOP_ARGS = 8   # iterator + [end, raw_indices, n_total - 1, buffer, all_pubkeys, message, merkle_chunks]
    loop_frame = f.alloc()
    p.hint_request_memory(loop_frame, K(Label("@loop_frame_size")))   # patched below: the loop's frame size is known after its body
    p.deref(loop_frame, 0, K(Label("after_loop")))
    p.deref(loop_frame, 1, FP(0))
    p.deref(loop_frame, 2, K(0))
    for k, cell in enumerate((meta + 2, raw_idx, ntm1, buffer, all_pk, msg_ptr, mchunks_ptr)):
        p.deref(loop_frame, 3 + k, M(cell))
    p.jump(K(1), K(Label("xmss_loop")), M(loop_frame))
    p.label("after_loop")
    p.add(M(meta + 2), K(0), M(n_total))           # counter == n_total (no recursions: counter = n_raw_xmss)
    # n_recursions == 0: the bytecode claim is (0^n_vars, bytecode[0]) (main.py:222-228)
    for k in range(L["n_vars"]):
        copy_5(M(ptr_plus(data, L["claim"] + DIM * k)), K(ZERO_VEC_PTR))
    for k in range(DIM):
        p.deref(data, L["claim"] + L["n_vars"] * DIM + k, K(0))     # BYTECODE_ZERO_EVAL = instructions_multilinear[0] = 0
    # slice_hash_with_iv(data_buf, n_chunks, pub_mem) (hashing.py:88-96): the last compression lands on the public input
    nch = L["n_chunks"]
    in_states = f.alloc((nch - 1) * DIGEST_LEN)
    p.poseidon16(K(ZERO_VEC_PTR), M(data), K(A(in_states)))
    for j in range(1, nch):
        dst = K(0) if j == nch - 1 else K(A(in_states + 8 * j))
        p.poseidon16(K(A(in_states + 8 * (j - 1))), M(ptr_plus(data, 8 * j)), dst)
    p.return_from_main(f.alloc())
    p.label("unsupported")
    p.panic()
    p.starting_frame_memory = f.top

    # ================================================================ the loop function ====================================================
    # frame: [return_pc, saved_fp, i, end, raw_indices, n_total - 1, buffer, all_pubkeys, message, merkle_chunks, locals ...]
    g = Frame(2 + LOOP_ARGS)
    I, END, RAW, NTM1, BUF, APK, MSG, MCH = 2, 3, 4, 5, 6, 7, 8, 9
    p.hint_parallel_batch_start(LOOP_ARGS, M(END))
    p.label("xmss_loop")
    d, dinv, nz, omnz = g.alloc(), g.alloc(), g.alloc(), g.alloc()
    p.add(M(d), M(END), M(I))                      # d = i - end
    p.hint_inverse(M(d), dinv)
    p.mul(M(d), M(dinv), M(nz))                    # nz = (i != end)
    p.add(M(omnz), M(nz), K(1))
    p.mul(M(omnz), M(d), K(0))
    p.jump(M(nz), K(Label("xmss_body")), FP(0))
    p.jump(K(1), M(0), M(1))                       # i == end: return
    p.label("xmss_body")

    def gptr(off):  # a cell holding the address fp + off
        c = g.alloc()
        p.add(K(0), FP(off), M(c))
        return c

    t_ptr, idx = g.alloc(), g.alloc()
    p.add(M(RAW), M(I), M(t_ptr))
    p.deref(t_ptr, 0, M(idx))                      # idx = raw_indices[i]
    p.range_check(idx, M(NTM1), g.alloc(3))        # assert idx < n_total
    b_ptr = g.alloc()
    p.add(M(BUF), M(idx), M(b_ptr))
    p.deref(b_ptr, 0, M(I))                        # buffer[idx] = i
    i8, pk = g.alloc(), g.alloc()
    p.mul(M(idx), K(PUB_KEY_SIZE), M(i8))
    p.add(M(APK), M(i8), M(pk))                    # pk = all_pubkeys + idx * PUB_KEY_SIZE
    # ---------------------------------------------------------------- xmss_verify(pk, message, merkle_chunks) --------------------------
    WOTS = g.alloc(WOTS_SIG_SIZE)
    p.hint_witness("wots", WOTS)
    pp = g.alloc()
    p.add(M(pk), K(XMSS_DIGEST_LEN), M(pp))        # public_param = pub_key + XMSS_DIGEST_LEN
    # 1) encode (xmss_aggregate.py:50-66)
    AIR, PRE, PPB, ENCFE = g.alloc(8), g.alloc(8), g.alloc(10), g.alloc(8)
    copy_5(M(gptr(WOTS)), FP(AIR))                 # copy_6(randomness, a_input_right)
    p.add(M(WOTS + 5), K(0), M(AIR + 5))
    twp = g.alloc()
    p.add(K(0), K(TW_ENC), M(twp))
    p.deref(twp, 0, M(AIR + 6))
    p.deref(twp, 1, M(AIR + 7))
    p.poseidon16(M(MSG), M(gptr(AIR)), FP(PRE))
    ppm1 = g.alloc()
    p.add(M(pk), K(XMSS_DIGEST_LEN - 1), M(ppm1))
    copy_5(M(ppm1), FP(PPB))                       # copy_5(public_param - 1, public_params_paded_buff)
    copy_5(M(gptr(PPB + 5)), K(ZERO_VEC_PTR))      # set_to_5_zeros(buff + 5)
    PPP = PPB + 1                                  # public_params_paded = [pp(4) | zeros(4)]
    p.poseidon16(FP(PRE), FP(PPP), FP(ENCFE))
    # decomposition into 6-bit chunks (two chain steps each), checked (xmss_aggregate.py:68-88)
    ENC = g.alloc(NUM_ENCODING_FE * 4)
    p.hint_decompose_bits_xmss(FP(ENC), FP(ENCFE), K(NUM_ENCODING_FE), K(6))
    for i in range(NUM_ENCODING_FE):
        for j in range(4):
            p.range_check(ENC + 4 * i + j, K(CHAIN_LENGTH ** 2 - 1), g.alloc(3))
        part = ENC + 4 * i
        for j in range(1, 4):
            t, s = g.alloc(), g.alloc()
            p.mul(M(ENC + 4 * i + j), K(64 ** j), M(t))
            p.add(M(part), M(t), M(s))
            part = s
        diff, rem = g.alloc(), g.alloc()
        p.add(M(diff), M(ENCFE + i), M(part))      # partial_sum - encoding_fe[i]
        p.mul(M(diff), K(127), M(rem))             # remaining_i: inv(2^24) = -127
        p.range_check(rem, K(126), g.alloc(3))     # assert remaining_i < 127
    # 2) chains: 21 pairs through 64-entry jump tables (xmss_aggregate.py:91-116, chain_hash_pair :150-180)
    WPK = g.alloc((V // 2) * WOTS_PK_PAIR_STRIDE)
    CHAIN_BLOCK = 2 * (CHAIN_LENGTH - 1) + 2       # 14 hashes + pair_sum + jump
    tables = []
    ts = None
    for i in range(V // 2):
        off, dest, psum, tmp_a, tmp_b = g.alloc(), g.alloc(), g.alloc(), g.alloc(), g.alloc()
        dig = g.alloc(2 * (CHAIN_LENGTH - 2) * XMSS_DIGEST_LEN)
        p.mul(M(ENC + i), K(CHAIN_BLOCK), M(off))
        p.add(M(off), K(Label(f"chain_table_{i}")), M(dest))
        p.jump(K(1), M(dest), FP(0))
        p.label(f"chain_after_{i}")
        tables.append(dict(i=i, psum=psum, tmp=(tmp_a, tmp_b), dig=dig))
        if ts is None:
            ts = psum
        else:
            s = g.alloc()
            p.add(M(ts), M(psum), M(s))
            ts = s
    p.add(M(ts), K(0), K(TARGET_SUM))              # assert target_sum == TARGET_SUM
    # 3) wots_pk_hash (xmss_aggregate.py:183-197)
    ST = g.alloc((V // 2 + 1) * DIGEST_LEN)
    p.poseidon16(M(pp), K(ZERO_VEC_PTR), FP(ST), left=TW_WOTS_PK)
    for i in range(V // 2):
        p.poseidon16(FP(ST + 8 * i), FP(WPK + WOTS_PK_PAIR_STRIDE * i + 1), FP(ST + 8 * (i + 1)))
    LEAF = ST + 8 * (V // 2)
    # 4) xmss_merkle_verify (xmss_aggregate.py:262-295): 8 chunks of 4 levels through 16-entry jump tables
    MST = g.alloc(DIM * N_MERKLE_CHUNKS)
    MERKLE_BLOCK = 2 + 4 + 1                       # copy (2), 4 hashes, jump
    mtables = []
    for j in range(N_MERKLE_CHUNKS):
        chunk, off, dest, tmp = g.alloc(), g.alloc(), g.alloc(), g.alloc()
        bufs = g.alloc(10 + 8 + 8 + 8)
        p.deref(MCH, j, M(chunk))
        p.mul(M(chunk), K(MERKLE_BLOCK), M(off))
        p.add(M(off), K(Label(f"merkle_table_{j}")), M(dest))
        p.jump(K(1), M(dest), FP(0))
        p.label(f"merkle_after_{j}")
        mtables.append(dict(j=j, tmp=tmp, bufs=bufs))
    # ---------------------------------------------------------------- next iteration ----------------------------------------------------
    NEXT, ip1 = g.alloc(), g.alloc()
    p.hint_request_memory(NEXT, K(Label("@loop_frame_size")))
    p.deref(NEXT, 0, M(0))
    p.deref(NEXT, 1, M(1))
    p.add(M(I), K(1), M(ip1))
    p.deref(NEXT, 2, M(ip1))
    for a in range(3, 2 + LOOP_ARGS):
        p.deref(NEXT, a, M(a))
    p.jump(K(1), K(Label("xmss_loop")), M(NEXT))
    loop_frame_size = g.top

    # ================================================================ jump tables =========================================================
    def chain_hashes(inp, n, out, tweaks, dig):  # chain_hash_pa (xmss_aggregate.py:121-147)
        start = CHAIN_LENGTH - 1 - n
        if n == 1:
            p.poseidon16(FP(inp), FP(PPP), FP(out), half=True, left=tweaks + start * TWEAK_LEN)
            return
        p.poseidon16(FP(inp), FP(PPP), FP(dig), half=True, left=tweaks + start * TWEAK_LEN)
        for jj in range(1, n - 1):
            p.poseidon16(FP(dig + 4 * (jj - 1)), FP(PPP), FP(dig + 4 * jj), half=True, left=tweaks + (start + jj) * TWEAK_LEN)
        p.poseidon16(FP(dig + 4 * (n - 2)), FP(PPP), FP(out), half=True, left=tweaks + (start + n - 1) * TWEAK_LEN)

    for tb in tables:
        i = tb["i"]
        in_a, in_b = WOTS + RANDOMNESS_LEN + 4 * (2 * i), WOTS + RANDOMNESS_LEN + 4 * (2 * i + 1)
        out_a = WPK + WOTS_PK_PAIR_STRIDE * i + 1
        out_b = out_a + XMSS_DIGEST_LEN
        tw_a = TW_CHAIN + (2 * i) * CHAIN_LENGTH * TWEAK_LEN
        tw_b = TW_CHAIN + (2 * i + 1) * CHAIN_LENGTH * TWEAK_LEN
        p.label(f"chain_table_{i}")
        for n in range(CHAIN_LENGTH ** 2):
            start = p.here()
            raw_a, raw_b = n % CHAIN_LENGTH, n // CHAIN_LENGTH
            na, nb = CHAIN_LENGTH - 1 - raw_a, CHAIN_LENGTH - 1 - raw_b
            if na == 0:
                p.add(K(0), FP(in_a - 1), M(tb["tmp"][0]))
                copy_5(M(tb["tmp"][0]), FP(out_a - 1))        # copy_5(input_a - 1, output_a - 1)
            else:
                chain_hashes(in_a, na, out_a, tw_a, tb["dig"])
            if nb == 0:
                p.add(K(0), FP(in_b), M(tb["tmp"][1]))
                copy_5(M(tb["tmp"][1]), FP(out_b))            # copy_5(input_b, output_b)
            else:
                chain_hashes(in_b, nb, out_b, tw_b, tb["dig"] + (CHAIN_LENGTH - 2) * XMSS_DIGEST_LEN)
            p.add(K(0), K(raw_a + raw_b), M(tb["psum"]))      # pair_sum_ptr[0] = raw_a + raw_b
            p.jump(K(1), K(Label(f"chain_after_{i}")), FP(0))
            assert p.here() - start <= CHAIN_BLOCK
            while p.here() - start < CHAIN_BLOCK:
                p.panic()
    for mt in mtables:
        j = mt["j"]
        buf0 = mt["bufs"] + 1
        buf = [buf0, mt["bufs"] + 10, mt["bufs"] + 18, mt["bufs"] + 26]
        state_in = LEAF if j == 0 else MST + 1 + DIM * (j - 1)
        state_out = M(pk) if j == N_MERKLE_CHUNKS - 1 else FP(MST + 1 + DIM * j)   # the last chunk writes onto the expected root
        p.label(f"merkle_table_{j}")
        for b in range(16):  # do_4_merkle_levels (xmss_aggregate.py:210-259)
            start = p.here()
            bit = [(b >> k) & 1 for k in range(4)]
            if bit[0]:
                p.add(K(0), FP(state_in - 1), M(mt["tmp"]))
                copy_5(M(mt["tmp"]), FP(buf0 - 1))            # state_in is the LEFT child
                sib = buf0 + XMSS_DIGEST_LEN
            else:
                p.add(K(0), FP(state_in), M(mt["tmp"]))
                copy_5(M(mt["tmp"]), FP(buf0 + XMSS_DIGEST_LEN))
                sib = buf0
            for lv in range(3):
                p.hint_witness("xmss_merkle_node", sib)       # the sibling of this level, before the hash that reads it
                nxt = buf[lv + 1]
                out, sib = (nxt, nxt + XMSS_DIGEST_LEN) if bit[lv + 1] else (nxt + XMSS_DIGEST_LEN, nxt)
                p.poseidon16(FP(PPP), FP(buf[lv]), FP(out), half=True, left=TW_MERKLE + (4 * j + lv) * TWEAK_LEN)
            p.hint_witness("xmss_merkle_node", sib)
            p.poseidon16(FP(PPP), FP(buf[3]), state_out, half=True, left=TW_MERKLE + (4 * j + 3) * TWEAK_LEN)
            p.jump(K(1), K(Label(f"merkle_after_{j}")), FP(0))
            assert p.here() - start == MERKLE_BLOCK
    p.labels["@loop_frame_size"] = loop_frame_size   # a "label" used as a plain constant by the two frame allocations
    bc = p.finalize(log_size if pad else None)
    bc.info = dict(layout=L, loop_frame_size=loop_frame_size, main_frame_size=f.top)
    return bc


# ================================================================================================================================================
# the witness of one run (type_1_aggregation.rs:206-380)
# ================================================================================================================================================
SNARK_DOMAIN_SEP = [130704175, 1303721200, 493664240, 1035493700, 2063844858, 1410214009, 1938905908, 1696767928]  # lean_prover/src/lib.rs:30-32


def compress_slice(x, data, use_iv):
    """poseidon_compress_slice (utils/src/poseidon.rs:41-67)"""
    data = np.asarray(data, dtype=np.uint32).reshape(-1, 8)
    if use_iv:
        h = np.zeros(8, dtype=np.uint32)
        rest = data
    else:
        h = x.compress(np.concatenate([data[0], data[1]])[None, :])[0, :8]
        rest = data[2:]
    for chunk in rest:
        h = x.compress(np.concatenate([h, chunk])[None, :])[0, :8]
    return h


def tweak_table(slot):
    """compute_tweak_table (type_1_aggregation.rs:125-152), Montgomery words"""
    tw = np.zeros((N_TWEAKS, 4), dtype=np.uint32)
    tw[0, :2] = to_monty(make_tweak(TWEAK_ENCODING, 0, slot))
    tw[1:1 + V * CHAIN_LENGTH, :2] = to_monty(make_tweak(TWEAK_CHAIN, np.arange(V * CHAIN_LENGTH), slot))
    tw[1 + V * CHAIN_LENGTH, :2] = to_monty(make_tweak(TWEAK_WOTS_PK, 0, slot))
    for level in range(LOG_LIFETIME):
        tw[2 + V * CHAIN_LENGTH + level, :2] = to_monty(make_tweak(TWEAK_MERKLE, level + 1, slot >> (level + 1)))
    out = np.zeros(TWEAK_TABLE_SIZE, dtype=np.uint32)
    out[:4 * N_TWEAKS] = tw.reshape(-1)
    return out


def build_witness(bc, n_sigs, rng, slot=0x00C0FFEE, xmss=None, sig=None, message=None):
    """aggregate_type_1 up to the prove_execution call: real signatures, sorted public keys, the public-input buffer and its digest,
    the hint streams.  -> (public_input (8 words), vm.Witness, info dict)"""
    x = xmss or Xmss()
    L = bc.info["layout"]
    if sig is None:
        message = rand_field(rng, 8)
        sig = x.keygen_and_sign(rng, n_sigs, message, slot, rand_field)
    pks = np.concatenate([sig["root"], sig["pp"]], axis=1)                 # XmssPublicKey::flaten: [merkle_root | public_param]
    order = np.lexsort(from_monty(pks).T[::-1])                             # raw_xmss.sort_by public key
    pks = pks[order]
    sig = {k: v[order] for k, v in sig.items()}
    tw = tweak_table(slot)
    tweaks_hash = compress_slice(x, tw, use_iv=False)
    pubkeys_hash = compress_slice(x, pks.reshape(-1), use_iv=True)
    data = np.zeros(L["size"], dtype=np.uint32)
    data[0], data[1] = to_monty(TYPE_1_FLAG), to_monty(n_sigs)
    # bytecode claim of a run without children: the point 0^n_vars and bytecode[0] = 0 (reduce_bytecode_claims with no claims)
    domsep = x.compress(np.concatenate([bc.hash(), to_monty(SNARK_DOMAIN_SEP)])[None, :])[0, :8]
    data[L["domsep"]:L["domsep"] + 8] = domsep
    data[L["pubkeys_hash"]:L["pubkeys_hash"] + 8] = pubkeys_hash
    data[L["message"]:L["message"] + 8] = message
    data[L["merkle_chunks"]:L["merkle_chunks"] + 8] = to_monty([(~(slot >> (4 * c))) & 0xF for c in range(N_MERKLE_CHUNKS)])
    data[L["tweaks_hash"]:L["tweaks_hash"] + 8] = tweaks_hash
    public_input = compress_slice(x, data, use_iv=True)
    wots = np.concatenate([sig["randomness"], sig["chain_tips"].reshape(n_sigs, 4 * V)], axis=1)
    hints = {
        "input_data_num_chunks": [to_monty([L["n_chunks"]])],
        "input_data": [data],
        "is_split": [to_monty([0])],
        "tweak_table": [tw],
        "meta": [to_monty([0, 0, n_sigs])],
        "pubkeys": [pks.reshape(-1)],
        "raw_indices": [to_monty(np.arange(n_sigs))],
        "aggregate_sizes": [np.zeros(0, dtype=np.uint32)],
        "wots": list(wots),
        "xmss_merkle_node": list(sig["merkle_proof"].reshape(n_sigs * LOG_LIFETIME, 4)),
    }
    return public_input, Witness(bc, PREAMBLE_MEMORY_LEN, hints), dict(sig=sig, message=message, slot=slot, input_data=data)
