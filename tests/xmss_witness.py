"""TEST INFRASTRUCTURE: a consistent leanVM execution witness whose Poseidon16 calls are the ACTUAL hash calls of verifying
real XMSS signatures (first step) — what `xmss --n-signatures N` feeds the prover, minus the
zkDSL compiler and VM (out of scope): the instruction stream is written out by hand, straight-line, following the
aggregation program crates/rec_aggregation/zkdsl_implem/xmss_aggregate.py:

  per signature (xmss_verify, :42-116)                                                 Poseidon16 rows   ExtensionOp rows
    encode      poseidon16_compress(message, [randomness | tweak]) ; (pre, [pp | 0])          2         copy_6 / copy_5 / zeros  3
    chains      42 chains, 7 - encoding[i] x poseidon16_compress_half_hardcoded_left        110         copy_5 per untouched chain
    wots pk     compress_hardcoded_left(pp, ZERO_VEC, states, tweak) + 21 x compress         22
    merkle      8 chunks x 4 levels of compress_half_hardcoded_left(pp, buf, out, tweak)     32         copy_5 per chunk          8
                                                                                         = 166 (TARGET_SUM fixes the 110)
  with the precompile variants, operand layouts (hardcoded-left tweak slots, half outputs, the [0 | tip_a | tip_b | 0] pairs of
  the WOTS public key, the shared 4-word tweak table) and write-once memory cells of that program; copy_5 is
  `dot_product_ee(src, ONE_EF_PTR, dst)` (zkdsl_implem/utils.py:359-361), i.e. one ExtensionOp row.
  The ~330 other cycles per signature of the real program (range checks of the encoding, target sum, match_range dispatch;
  SURVEY.md §8: 500-550 cycles per signature) are modelled by `n_arith` ADD / MUL / DEREF instructions on cells of the
  signature's own memory region — real data flow and a realistic, scattered access pattern, not the program's exact arithmetic.
Differences from the real run that matter to the prover: the program is straight-line (no fp-relative loop), so the bytecode
table has one row per cycle (bytecode_acc is 0/1 instead of N per row) — table shapes, lookups, bus and AIR are the real ones.
The three AIRs, all lookups and the bus accept the witness (the oracle's and the library's verify_execution check it)."""
import numpy as np

from tests import oracle_binding as ob
from tests.oracle_binding import P
from tests.synth_witness import ONE, access_counters, ef_mul_vec
from tests.xmss_py import (CHAIN_LENGTH, LOG_LIFETIME, TWEAK_CHAIN, TWEAK_ENCODING, TWEAK_MERKLE, TWEAK_WOTS_PK, V, Xmss, make_tweak)

# precompile_data of the Poseidon16 table (lean_vm/src/tables/poseidon_16/mod.rs:94-98)
PD_BASE, PD_HALF, PD_LEFT, PD_OFFSET = 1, 4, 8, 16


def build(orc, rng, n_sigs, slot=0x00C0FFEE, n_arith=330, log_exec=None, log_pos=None, log_ext=None, log_memory=None,
          log_bytecode=None, compress=None, fill_rows=None):
    """-> witness dict in the format of tests/synth_witness.build (+ 'xmss': the signatures)."""
    M = lambda x: orc.to_monty(np.asarray(x, dtype=np.uint64))  # noqa: E731
    xm = Xmss(orc)
    if compress is not None:
        xm.compress = compress
    message = ob.rand_field(rng, 8)
    sig = xm.keygen_and_sign(rng, n_sigs, message, slot, ob.rand_field)
    enc = sig["encoding"]                         # (n, V): chain i needs 7 - enc[i] hashes, starting at step enc[i]
    n_hash = (CHAIN_LENGTH - 1) - enc
    S = n_sigs
    # ---------------------------------------------------------------- memory map ------------------------------------------
    n_pub = 16
    Z, ONE_EF, NULL, MSG, TWK = 64, 80, 96, 112, 128     # zero vector (16), [1,0,0,0,0], null hash, message (8), tweak table
    n_tweaks = 1 + V * CHAIN_LENGTH + 1 + LOG_LIFETIME
    TW_ENC, TW_CHAIN = TWK, TWK + 4
    TW_PK = TW_CHAIN + V * CHAIN_LENGTH * 4
    TW_MERKLE = TW_PK + 4
    base = TWK + ((4 * n_tweaks + 15) // 16) * 16
    # per-signature region (offsets relative to the region start)
    o = {}
    cur = 0
    for name, size in (("PK", 8), ("WOTS", 6 + 4 * V), ("AIR", 8), ("PRE", 8), ("PPB", 10), ("ENCFE", 8), ("WPK", 10 * (V // 2)),
                       ("DIG", 4 * (110 - 1)), ("STATES", 8 * (V // 2 + 1)), ("MBUF", 8 * (10 + 8 + 8 + 8)), ("MST", 5 * 8 + 1),
                       ("AR", 2 * n_arith + 8), ("PAD", 16)):
        o[name] = cur
        cur += size
    REGION = ((cur + 15) // 16) * 16
    need = base + S * REGION + 32
    log_memory = log_memory or max(16, int(np.ceil(np.log2(need))))
    mem_len = 1 << log_memory
    assert need <= mem_len
    memory = np.zeros(mem_len, dtype=np.uint32)
    public_input = ob.rand_field(rng, n_pub)
    memory[:n_pub] = public_input
    memory[ONE_EF] = ONE
    memory[NULL:NULL + 8] = orc.poseidon16_compress(np.zeros((1, 16), dtype=np.uint32))[0][:8]
    memory[MSG:MSG + 8] = message
    # tweak table: 4-word slots [tw0, tw1, 0, 0] (xmss_aggregate.py:24-32)
    tw = np.zeros((n_tweaks, 4), dtype=np.uint32)
    tw[0, :2] = M(make_tweak(TWEAK_ENCODING, 0, slot))
    tw[1:1 + V * CHAIN_LENGTH, :2] = M(make_tweak(TWEAK_CHAIN, np.arange(V * CHAIN_LENGTH), slot))
    tw[1 + V * CHAIN_LENGTH, :2] = M(make_tweak(TWEAK_WOTS_PK, 0, slot))
    for level in range(LOG_LIFETIME):
        tw[2 + V * CHAIN_LENGTH + level, :2] = M(make_tweak(TWEAK_MERKLE, level + 1, slot >> (level + 1)))
    memory[TWK:TWK + 4 * n_tweaks] = tw.reshape(-1)
    R = base + REGION * np.arange(S)              # region start per signature
    A = lambda name, extra=0: R + o[name] + extra  # noqa: E731 — absolute address per signature

    def put(addr, vals):  # memory[addr[s] + j] = vals[s, j]
        vals = np.asarray(vals, dtype=np.uint32)
        memory[(addr[:, None] + np.arange(vals.shape[1])[None, :]).reshape(-1)] = vals.reshape(-1)

    def get(addr, n):
        return memory[addr[:, None] + np.arange(n)[None, :]]

    put(A("PK"), np.concatenate([sig["root"], sig["pp"]], axis=1))
    put(A("WOTS"), np.concatenate([sig["randomness"], sig["chain_tips"].reshape(S, 4 * V)], axis=1))

    # ---------------------------------------------------------------- instruction stream --------------------------------
    # Every instruction is recorded as (sig index array, kind, fields...) in program order per signature; signatures are laid
    # out one after the other.  kind: 'pos' (poseidon call), 'ext' (copy_5), 'ar' (ADD / MUL / DEREF filler)
    pos_calls, ext_calls = [], []   # lists of dicts of per-signature arrays (+ 'order': position inside the signature's code)
    order = [0]

    def next_order():
        order[0] += 1
        return order[0]

    def poseidon(sel, arg_a, arg_b, res, half=False, left_off=None):
        """record + execute one precompile call for the signatures `sel` (Poseidon16Precompile::execute, mod.rs:209-289)"""
        arg_a, arg_b, res = np.asarray(arg_a), np.asarray(arg_b), np.asarray(res)
        hard = left_off is not None
        first = np.asarray(left_off) if hard else arg_a
        second = arg_a if hard else arg_a + 4
        x = np.concatenate([get(first, 4), get(second, 4), get(arg_b, 8)], axis=1)
        out = xm.compress(x)
        put(res, out[:, :4] if half else out[:, :8])
        pos_calls.append(dict(sel=sel, a=arg_a, b=arg_b, r=res, half=np.full(sel.size, half), hard=np.full(sel.size, hard),
                              off=(np.asarray(left_off) if hard else np.zeros(sel.size, dtype=np.int64)), first=first, second=second,
                              inputs=x, order=next_order()))

    def copy5(sel, src, dst):
        """copy_5 = dot_product_ee(src, ONE_EF_PTR, dst): dst[0..5) = src[0..5) * 1 (one ExtensionOp row, mode mul / ee / len 1)"""
        src, dst = np.asarray(src), np.asarray(dst)
        put(dst, get(src, 5))
        ext_calls.append(dict(sel=sel, a=src, r=dst, order=next_order()))

    allS = np.arange(S)
    # 1) encode (xmss_aggregate.py:50-66)
    copy5(allS, A("WOTS"), A("AIR"))                                   # copy_6: dot product + one cell
    put(A("AIR", 5), get(A("WOTS", 5), 1))
    put(A("AIR", 6), np.tile(tw[0, :2], (S, 1)))                       # a_input_right[6..8] = encoding tweak
    poseidon(allS, np.full(S, MSG), A("AIR"), A("PRE"))
    copy5(allS, A("PK", 3), A("PPB"))                                  # copy_5(public_param - 1, buff): [root[3] | pp(4)]
    copy5(allS, np.full(S, Z), A("PPB", 5))                            # set_to_5_zeros
    poseidon(allS, A("PRE"), A("PPB", 1), A("ENCFE"))
    assert np.array_equal(get(A("ENCFE"), 8), xm.encode(message, slot, sig["pp"], sig["randomness"])[3])
    # 2) chains (:87-116, chain_hash_pa :121-147): digests of one signature are packed one after the other
    dig_used = np.zeros(S, dtype=np.int64)
    for c in range(V):
        pair, side = divmod(c, 2)
        src = A("WOTS", 6 + 4 * c)                                     # chain_start
        dst = A("WPK", 10 * pair + 1 + 4 * side)                       # chain_end inside [0 | tip_a | tip_b | 0]
        n = n_hash[:, c]
        none = np.nonzero(n == 0)[0]
        if none.size:                                                  # copy_5(input_a - 1, output_a - 1) / copy_5(input_b, output_b)
            copy5(none, src[none] - (1 - side), dst[none] - (1 - side))
        cur_in = src.copy()
        for k in range(CHAIN_LENGTH - 1):
            act = np.nonzero(n > k)[0]
            if not act.size:
                break
            last = n[act] == k + 1
            out = np.where(last, dst[act], R[act] + o["DIG"] + 4 * dig_used[act])
            dig_used[act] += ~last
            twa = TW_CHAIN + 4 * (c * CHAIN_LENGTH + enc[act, c] + k)
            poseidon(act, cur_in[act], R[act] + o["PPB"] + 1, out, half=True, left_off=twa)
            cur_in[act] = out
    assert dig_used.max() <= 109
    # 3) WOTS public key hash (wots_pk_hash :183-197)
    poseidon(allS, A("PK", 4), np.full(S, Z), A("STATES"), left_off=np.full(S, TW_PK))
    for i in range(V // 2):
        poseidon(allS, A("STATES", 8 * i), A("WPK", 10 * i + 1), A("STATES", 8 * (i + 1)))
    # 4) Merkle path (do_4_merkle_levels :210-259, xmss_merkle_verify :262-295): the current node sits in the left or right half
    #    of an 8-word buffer next to the hinted sibling; after a chunk of 4 levels the node is a 4-word state
    node = A("STATES", 8 * (V // 2))                                   # merkle_leaf = first 4 words of the last state
    for ch in range(LOG_LIFETIME // 4):
        bufs = [A("MBUF", 34 * ch + 1), A("MBUF", 34 * ch + 10), A("MBUF", 34 * ch + 18), A("MBUF", 34 * ch + 26)]
        for lv in range(4):
            level = 4 * ch + lv
            is_left = ((slot >> level) & 1) == 0
            buf = bufs[lv]
            sib = sig["merkle_proof"][:, level]
            if lv == 0:                                                # the incoming state is copied next to the hinted sibling
                if is_left:
                    copy5(allS, node - 1, buf - 1)                     # copy_5(state_in - 1, buf0 - 1)
                    put(buf + 4, sib)
                else:
                    put(buf, sib)
                    copy5(allS, node, buf + 4)                         # copy_5(state_in, buf0 + 4)
            else:
                put(buf + (4 if is_left else 0), sib)                  # hint_witness("xmss_merkle_node", ..)
            if lv < 3:
                nxt_left = ((slot >> (level + 1)) & 1) == 0
                out = bufs[lv + 1] + (0 if nxt_left else 4)
            else:
                out = A("PK") if ch == LOG_LIFETIME // 4 - 1 else A("MST", 5 * ch + 1)   # last chunk writes onto the expected root
            if ch == LOG_LIFETIME // 4 - 1 and lv == 3:
                # the root cells are already written (public key): write-once memory makes the store an equality check
                x = np.concatenate([tw[2 + V * CHAIN_LENGTH + level][None, :].repeat(S, 0), sig["pp"], get(buf, 8)], axis=1)
                assert np.array_equal(xm.compress(x)[:, :4], sig["root"]), "Merkle path does not lead to the public key"
            poseidon(allS, A("PK", 4), buf, out, half=True, left_off=np.full(S, TW_MERKLE + 4 * level))
            node = out
    n_pos_per_sig = np.zeros(S, dtype=np.int64)
    for c_ in pos_calls:
        np.add.at(n_pos_per_sig, c_["sel"], 1)
    assert np.all(n_pos_per_sig == 166), n_pos_per_sig[:4]
    # 5) filler arithmetic on the signature's own cells: x_j = a op b with a, b among the encoding words and earlier results
    ar_rows = []
    if n_arith:
        cells = np.zeros((S, 8 + n_arith), dtype=np.uint32)
        cells[:, :8] = get(A("ENCFE"), 8)
        cell_addr = np.concatenate([A("ENCFE")[:, None] + np.arange(8), (A("AR")[:, None] + 2 * np.arange(n_arith))], axis=1)
        jr = np.random.default_rng(12345)
        for j in range(n_arith):
            kind = j % 3                                               # 0 ADD, 1 MUL, 2 DEREF
            ia, ic = int(jr.integers(max(0, j - 40), 8 + j)), int(jr.integers(0, 8 + j))
            xa, xc = cells[:, ia], cells[:, ic]
            new = A("AR", 2 * j)
            if kind == 0:
                val = ((xa.astype(np.uint64) + xc) % P).astype(np.uint32)
            elif kind == 1:
                from tests.synth_witness import mmul
                val = mmul(xa, xc)
            else:
                val = xc                                               # m[m[ptr] + 7] = m[c]: the pointer cell sits next to the result
                memory[new + 1] = M(new - 7)
            memory[new] = val
            cells[:, 8 + j] = val
            ar_rows.append(dict(kind=kind, a=(new + 1) if kind == 2 else cell_addr[:, ia], b=new, c=cell_addr[:, ic], order=next_order()))

    # ---------------------------------------------------------------- program order -> pc ---------------------------------
    # per signature: instructions sorted by `order`; a signature's block follows the previous one's
    ev = []   # (sig, order, type, index into list, row inside)
    for li, c_ in enumerate(pos_calls):
        ev.append(np.stack([c_["sel"], np.full(c_["sel"].size, c_["order"]), np.zeros(c_["sel"].size, dtype=np.int64),
                            np.full(c_["sel"].size, li), np.arange(c_["sel"].size)], axis=1))
    for li, c_ in enumerate(ext_calls):
        ev.append(np.stack([c_["sel"], np.full(c_["sel"].size, c_["order"]), np.ones(c_["sel"].size, dtype=np.int64),
                            np.full(c_["sel"].size, li), np.arange(c_["sel"].size)], axis=1))
    for li, c_ in enumerate(ar_rows):
        ev.append(np.stack([allS, np.full(S, c_["order"]), np.full(S, 2), np.full(S, li), allS], axis=1))
    ev = np.concatenate(ev, axis=0)
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    n_instr = ev.shape[0]
    ending_pc = n_instr
    log_exec = log_exec or max(8, int(np.ceil(np.log2(n_instr + 1))))
    log_bytecode = log_bytecode or log_exec
    n_exec = 1 << log_exec
    assert n_instr < n_exec and n_instr < (1 << log_bytecode)
    pc_of = np.arange(n_instr)
    bytecode = np.zeros((1 << log_bytecode, 16), dtype=np.uint32)
    ex = np.zeros((24, n_exec), dtype=np.uint32)
    ex[2:5] = int(M(Z))
    is_pos, is_ext, is_ar = ev[:, 2] == 0, ev[:, 2] == 1, ev[:, 2] == 2

    # flat per-call arrays in list order, then indexed by (list, row)
    def flat(calls, key):
        off = np.cumsum([0] + [c_["sel"].size if "sel" in c_ else S for c_ in calls])
        return np.concatenate([np.asarray(c_[key]) for c_ in calls]), off

    # ---- poseidon calls
    pa, poff = flat(pos_calls, "a")
    pb, _ = flat(pos_calls, "b")
    pr, _ = flat(pos_calls, "r")
    phalf, _ = flat(pos_calls, "half")
    phard, _ = flat(pos_calls, "hard")
    poffs, _ = flat(pos_calls, "off")
    pfirst, _ = flat(pos_calls, "first")
    psecond, _ = flat(pos_calls, "second")
    pin = np.concatenate([c_["inputs"] for c_ in pos_calls], axis=0)
    pidx = poff[ev[is_pos, 3]] + ev[is_pos, 4]                         # program order -> flat call index
    pdata = PD_BASE + PD_HALF * phalf + PD_LEFT * phard + PD_OFFSET * poffs * phard
    pcs_pos = pc_of[is_pos]
    bytecode[pcs_pos, 0], bytecode[pcs_pos, 1], bytecode[pcs_pos, 2] = M(pa[pidx]), M(pb[pidx]), M(pr[pidx])
    bytecode[pcs_pos, 3:6] = ONE
    bytecode[pcs_pos, 11] = M(pdata[pidx])
    # ---- extension-op calls (copy_5)
    if ext_calls:
        ea, eoff = flat(ext_calls, "a")
        er, _ = flat(ext_calls, "r")
        eidx = eoff[ev[is_ext, 3]] + ev[is_ext, 4]
        pcs_ext = pc_of[is_ext]
        EXT_AUX = 16 + 64                                              # mode mul, ee, len 1 (extension_op/mod.rs:10-14)
        bytecode[pcs_ext, 0], bytecode[pcs_ext, 1], bytecode[pcs_ext, 2] = M(ea[eidx]), int(M(ONE_EF)), M(er[eidx])
        bytecode[pcs_ext, 3:6] = ONE
        bytecode[pcs_ext, 11] = int(M(EXT_AUX))
    # ---- arithmetic
    if ar_rows:
        aa = np.stack([np.broadcast_to(c_["a"], (S,)) for c_ in ar_rows])   # (n_arith, S)
        ab = np.stack([c_["b"] for c_ in ar_rows])
        ac = np.stack([np.broadcast_to(c_["c"], (S,)) for c_ in ar_rows])
        akind = np.array([c_["kind"] for c_ in ar_rows])
        li, si = ev[is_ar, 3], ev[is_ar, 4]
        pcs_ar = pc_of[is_ar]
        k_ = akind[li]
        a_addr, b_addr, c_addr = aa[li, si], ab[li, si], ac[li, si]
        bytecode[pcs_ar, 0] = M(a_addr)
        bytecode[pcs_ar, 1] = M(np.where(k_ == 2, 7, b_addr))
        bytecode[pcs_ar, 2] = M(c_addr)
        bytecode[pcs_ar, 4] = np.where(k_ == 2, ONE, 0)
        bytecode[pcs_ar, 8] = np.where(k_ == 1, ONE, 0)
        bytecode[pcs_ar, 10] = M(np.where(k_ == 0, 1, np.where(k_ == 2, 2, 0)))
    bytecode[ending_pc, :12] = [ONE, int(M(ending_pc)), 0, ONE, ONE, 0, ONE, 0, 0, ONE, 0, 0]
    # ---- execution table
    pcs = np.minimum(np.arange(n_exec), ending_pc)
    ex[0] = M(pcs)
    ex[8:20] = bytecode[pcs, :12].T
    pre = np.nonzero(is_pos | is_ext)[0]
    ex[20, pre] = ONE
    ex[21, pre], ex[22, pre], ex[23, pre] = ex[8, pre], ex[9, pre], ex[10, pre]   # immediates
    ex[21, n_instr:] = ONE
    ex[22, n_instr:] = int(M(ending_pc))
    if ar_rows:
        ex[2, pcs_ar], ex[3, pcs_ar], ex[4, pcs_ar] = M(a_addr), M(b_addr), M(c_addr)
        ex[5, pcs_ar], ex[6, pcs_ar], ex[7, pcs_ar] = memory[a_addr], memory[b_addr], memory[c_addr]
        ex[21, pcs_ar] = memory[a_addr]
        ex[22, pcs_ar] = np.where(k_ == 2, int(M(7)), memory[b_addr])
        ex[23, pcs_ar] = memory[c_addr]
    # ---- poseidon table
    n_calls = int(is_pos.sum())
    log_pos = log_pos or max(8, int(np.ceil(np.log2(n_calls + 1))))
    n_pos = 1 << log_pos
    assert n_calls <= n_pos
    rows = np.zeros((n_pos, 109), dtype=np.uint32)
    rows[:, 6], rows[:, 7] = int(M(Z)), int(M(Z + 4))
    rows[:, 1], rows[:, 2] = int(M(Z)), int(M(NULL))
    rows[:n_calls, 0] = ONE
    rows[:n_calls, 1], rows[:n_calls, 2] = M(pb[pidx]), M(pr[pidx])
    rows[:n_calls, 3] = np.where(phalf[pidx], ONE, 0)
    rows[:n_calls, 4] = np.where(phard[pidx], ONE, 0)
    rows[:n_calls, 5] = M(poffs[pidx] * phard[pidx])
    rows[:n_calls, 6], rows[:n_calls, 7] = M(pfirst[pidx]), M(psecond[pidx])
    rows[:n_calls, 9:25] = pin[pidx]
    rows = np.ascontiguousarray(rows)
    if fill_rows is None:
        import ctypes
        orc.lib.orc_poseidon16_fill_rows(rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n_pos))
    else:
        fill_rows(rows)
    # trace_gen.rs:118-147: the unconstrained output columns of compression rows hold what their lookup reads from memory
    res = pr[pidx]
    rows[:n_calls, 101:109] = memory[res[:, None] + 8 + np.arange(8)[None, :]]
    hf = np.nonzero(phalf[pidx])[0]
    rows[hf, 97:101] = memory[res[hf, None] + 4 + np.arange(4)[None, :]]
    pos = np.zeros((111, n_pos), dtype=np.uint32)
    pos[:109] = rows.T
    pos[109] = int(M(Z))
    pos[109, :n_calls] = M(pa[pidx])
    pos[110] = ONE
    pos[110, :n_calls] = M(pdata[pidx])
    # ---- extension_op table
    n_ext_calls = int(is_ext.sum())
    log_ext = log_ext or max(8, int(np.ceil(np.log2(n_ext_calls + 2))))
    n_ext = 1 << log_ext
    assert n_ext_calls < n_ext
    ext = np.zeros((31, n_ext), dtype=np.uint32)
    ext[1] = ONE
    ext[2] = ONE
    ext[30] = int(M(64))
    ext[6] = ext[7] = ext[13] = int(M(Z))
    if ext_calls:
        src, dst = ea[eidx], er[eidx]
        va = memory[src[:, None] + np.arange(5)[None, :]]
        one = np.zeros((n_ext_calls, 5), dtype=np.uint32)
        one[:, 0] = ONE
        prod = ef_mul_vec(va, one)
        assert np.array_equal(prod, memory[dst[:, None] + np.arange(5)[None, :]])
        e = np.zeros((31, n_ext_calls), dtype=np.uint32)
        e[1] = ONE                      # start
        e[2] = ONE                      # len = 1
        e[4] = ONE                      # flag_mul
        e[6], e[7], e[13] = M(src), int(M(ONE_EF)), M(dst)
        e[8:13] = prod.T                # computation
        e[14:19] = va.T
        e[19:24] = one.T
        e[24:29] = prod.T
        e[29] = ONE
        e[30] = int(M(EXT_AUX))
        ext[:, :n_ext_calls] = e
    tables = {0: ex, 1: ext, 2: pos}
    memory_acc, bytecode_acc = access_counters(orc, tables, mem_len, 1 << log_bytecode)
    return dict(log_inv_rate=1, log_memory=log_memory, log_bytecode=log_bytecode, ending_pc=ending_pc, public_memory_size=n_pub,
                public_input=public_input, bytecode_hash=ob.rand_field(rng, 8), bytecode=np.ascontiguousarray(bytecode),
                bytecode_acc=bytecode_acc, memory=memory, memory_acc=memory_acc, tables=tables,
                log_rows={0: log_exec, 1: log_ext, 2: log_pos}, xmss=sig, n_sigs=n_sigs,
                counts=dict(poseidon=n_calls, extension_op=n_ext_calls, cycles=n_instr))
