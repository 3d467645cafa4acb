"""Synthetic but CONSISTENT execution witness for the proving slice after witness generation (TEST INFRASTRUCTURE).

The reference obtains (memory, tables) by running the zkDSL aggregation program in its VM; that interpreter is out of
scope.  This module builds by hand a tiny straight-line leanVM program that the three AIRs, the memory /
bytecode lookups and the precompile bus all accept, so that prove_execution's slice can be proven AND verified:

  pc = 0 .. N-1 : `poseidon16_compress(a_i, b_i) -> res_i` precompile calls with immediate operands
                  (execution/air.rs:56-129: flag_a = flag_b = flag_c = 1, aux = mul = jump = 0  =>  is_precompile = 1,
                   nu_a/b/c = operand_a/b/c, next pc = pc + 1, fp unchanged)
  pc = N        : the self-loop jump used as padding row (execution/mod.rs:59-74)
  optionally, after the Poseidon calls (still straight-line, fp = 0):
    `n_arith` instructions cycling ADD (aux = 1: m[b] = m[a] + m[c]), MUL (mul = 1: m[b] = m[a] * m[c]) and DEREF
       (aux = 2: m[m[a] + operand_b] = m[c]) with memory operands (flags 0), execution/air.rs:96-112;
    `ext_calls` = [(op, is_be, size, count)]: extension-op precompile calls (op in add / mul / poly_eq), each `size`
       rows of the ExtensionOp table written as exec_multi_row does (extension_op/exec.rs:95-189: element values,
       backward accumulation, result to memory, len counting down, aux = mode bits + 64 * len).
Poseidon16 table: one active row per call + padding rows (poseidon_16/mod.rs:176-199); extension_op table: the rows of
`ext_calls` then padding rows (extension_op/mod.rs:125-135).  Memory holds the inputs, outputs, the zero vector and
the null hash.  The oracle proving AND verifying such a witness pins its restatement of the three AIRs against the
reference's executor semantics (a wrong constraint would reject a trace the reference's VM produces).
"""
import numpy as np

from tests import oracle_binding as ob
from tests.oracle_binding import P

ONE = 0x01FFFFFE


RINV = pow(1 << 32, -1, P)


def mmul(a, b):
    """Montgomery product of uint32 arrays (numpy twin of monty_reduce, monty_31/utils.rs:107-127)."""
    t = (np.asarray(a, dtype=np.uint64) * np.asarray(b, dtype=np.uint64)) % P
    return ((t * RINV) % P).astype(np.uint32)


def ef_mul_vec(a, b):
    """(..., 5) x (..., 5) product mod X^5 + X^2 - 1 (quintic_extension/extension.rs:531-548), Montgomery coordinates."""
    d = np.zeros(a.shape[:-1] + (9,), dtype=np.uint64)
    for i in range(5):
        for j in range(5):
            d[..., i + j] += mmul(a[..., i], b[..., j])
    d %= P
    c = d[..., :5].copy()
    neg = lambda x: (P - x) % P  # noqa: E731
    c[..., 0] += d[..., 5] + neg(d[..., 8])            # X^5 = 1 - X^2, X^8 = X^3 + X^2 - 1
    c[..., 1] += d[..., 6]                              # X^6 = X - X^3
    c[..., 2] += neg(d[..., 5]) + d[..., 7] + d[..., 8]  # X^7 = X^2 - X^4
    c[..., 3] += neg(d[..., 6]) + d[..., 8]
    c[..., 4] += neg(d[..., 7])
    return (c % P).astype(np.uint32)


EXT_FLAG = {"add": 8, "mul": 16, "poly_eq": 32}  # extension_op/mod.rs:10-14
EXT_IS_BE, EXT_LEN_MULT = 4, 64


def _ext_rows(orc, M, memory, op, is_be, size, count, ptr_a0, ptr_b0, ptr_r0):
    """`count` calls of one mode laid out back to back; returns (rows (31, count*size), per-call (ptr_a, ptr_b, ptr_res, aux))
    and writes the results into memory.  Vectorised over the calls; follows exec_multi_row (extension_op/exec.rs:95-189)."""
    a_stride = 1 if is_be else 5
    i = np.arange(size)
    call = np.arange(count)
    ptr_a = ptr_a0 + call * size * a_stride
    ptr_b = ptr_b0 + call * size * 5
    ptr_r = ptr_r0 + call * 5
    idx_a = ptr_a[:, None] + i[None, :] * a_stride          # (count, size)
    idx_b = ptr_b[:, None] + i[None, :] * 5
    va_mem = memory[idx_a[..., None] + np.arange(5)]         # the 5 gathered words (fill_trace_extension_op)
    vb = memory[idx_b[..., None] + np.arange(5)]
    va = va_mem.copy()
    if is_be:
        va[..., 1:] = 0                                      # EF::from(base)
    add = lambda x, y: ((x.astype(np.uint64) + y) % P).astype(np.uint32)          # noqa: E731
    sub = lambda x, y: ((x.astype(np.uint64) + P - y) % P).astype(np.uint32)      # noqa: E731
    mul = ef_mul_vec
    assert np.array_equal(mul(va[0, 0], vb[0, 0]), orc.ef_mul(va[0, 0], vb[0, 0]))  # the numpy twin agrees with the oracle
    one = np.zeros(5, dtype=np.uint32)
    one[0] = ONE
    if op == "add":
        elem = add(va, vb)
    elif op == "mul":
        elem = mul(va, vb)
    else:
        ab = mul(va, vb)
        elem = add(sub(sub(add(ab, ab), va), vb), one)       # 2ab - a - b + 1
    comp = np.zeros_like(elem)
    comp[:, size - 1] = elem[:, size - 1]
    for k in range(size - 2, -1, -1):
        comp[:, k] = mul(elem[:, k], comp[:, k + 1]) if op == "poly_eq" else add(elem[:, k], comp[:, k + 1])
    res = comp[:, 0]
    memory[ptr_r[:, None] + np.arange(5)] = res
    n = count * size
    rows = np.zeros((31, n), dtype=np.uint32)
    mode_bits = EXT_FLAG[op] + (EXT_IS_BE if is_be else 0)
    cur_len = (size - i)[None, :].repeat(count, 0)
    rows[0] = ONE if is_be else 0
    rows[1] = np.where(i == 0, ONE, 0)[None, :].repeat(count, 0).reshape(-1)
    rows[2] = M(cur_len.reshape(-1))
    rows[3 + ["add", "mul", "poly_eq"].index(op)] = ONE
    rows[6], rows[7] = M(idx_a.reshape(-1)), M(idx_b.reshape(-1))
    rows[8:13] = comp.reshape(n, 5).T
    rows[13] = M(ptr_r[:, None].repeat(size, 1).reshape(-1))
    rows[14:19] = va_mem.reshape(n, 5).T
    rows[19:24] = vb.reshape(n, 5).T
    rows[24:29] = res[:, None, :].repeat(size, 1).reshape(n, 5).T
    rows[29] = rows[1]
    rows[30] = M((mode_bits + EXT_LEN_MULT * cur_len).reshape(-1))
    return rows, (ptr_a, ptr_b, ptr_r, np.full(count, mode_bits + EXT_LEN_MULT * size))


def build(orc, rng, n_calls, n_blocks=8, log_exec=8, log_pos=8, log_ext=8, log_memory=16, log_bytecode=8, fill_rows=None,
          n_arith=0, ext_calls=()):
    """fill_rows(rows) may replace the oracle's Poseidon trace generator for big tables (rows: (n, 109) uint32, in place)."""
    M = lambda x: orc.to_monty(np.asarray(x, dtype=np.uint64))  # noqa: E731
    n_exec, n_pos, n_ext = 1 << log_exec, 1 << log_pos, 1 << log_ext
    n_ext_calls = sum(c for _, _, _, c in ext_calls)
    n_instr = n_calls + n_arith + n_ext_calls
    assert n_instr < n_exec and n_calls <= n_pos and n_instr < (1 << log_bytecode)
    mem_len = 1 << log_memory
    memory = np.zeros(mem_len, dtype=np.uint32)
    # public input occupies the start of memory (public memory = public input padded to a power of two)
    n_pub = 16
    public_input = ob.rand_field(rng, n_pub)
    memory[:n_pub] = public_input
    Z = 64            # zero vector (16 zero words)
    NULL = 96         # null hash: compress(0^16)[0..8] followed by 8 zeros
    memory[NULL:NULL + 8] = orc.poseidon16_compress(np.zeros(16, dtype=np.uint32))[0][:8]
    base = 128
    assert base + 32 * n_blocks <= mem_len
    inputs = ob.rand_field(rng, (n_blocks, 16))
    outs = orc.poseidon16_compress(inputs)[:, :8]
    blk = memory[base:base + 32 * n_blocks].reshape(n_blocks, 32)
    blk[:, :16] = inputs
    blk[:, 16:24] = outs     # res: outputs_left; res+8..16 stay 0 (outputs_right = 0 in compress mode)
    call_blk = np.arange(n_calls) % n_blocks
    addr_a = base + 32 * call_blk
    addr_b, addr_r = addr_a + 8, addr_a + 16
    ending_pc = n_instr
    # ---- bytecode (row-major, stride 16; columns = the 12 instruction columns of the execution table) ----------
    # operand_a, operand_b, operand_c, flag_a, flag_b, flag_c, flag_c_fp, flag_ab_fp, mul, jump, aux, precompile_data
    bytecode = np.zeros((1 << log_bytecode, 16), dtype=np.uint32)
    bytecode[:n_calls, 0], bytecode[:n_calls, 1], bytecode[:n_calls, 2] = M(addr_a), M(addr_b), M(addr_r)
    bytecode[:n_calls, 3:6] = ONE
    bytecode[:n_calls, 11] = ONE
    bytecode[ending_pc, :12] = [ONE, int(M(ending_pc)), 0, ONE, ONE, 0, ONE, 0, 0, ONE, 0, 0]
    # ---- optional arithmetic instructions on memory operands: 4 cells (x, y, z, pointer) each --------------------------
    top = base + 32 * n_blocks
    ar = top + 4 * np.arange(n_arith)
    top += 4 * n_arith
    assert top + 8 <= mem_len
    kind = np.arange(n_arith) % 3          # 0 ADD, 1 MUL, 2 DEREF
    if n_arith:
        x, z = ob.rand_field(rng, n_arith), ob.rand_field(rng, n_arith)
        y = np.where(kind == 0, ((x.astype(np.uint64) + z) % P).astype(np.uint32), mmul(x, z))
        memory[ar], memory[ar + 2] = x, z
        memory[ar + 1] = np.where(kind == 2, z, y)                  # DEREF: the cell pointed at holds m[c]
        memory[ar + 3] = M(ar - 6)                                  # pointer: m[a] + operand_b = (ar - 6) + 7 = ar + 1
        pcs_ar = n_calls + np.arange(n_arith)
        # ADD / MUL: a = &x, b = &y, c = &z ; DEREF: a = &pointer, operand_b = 7 (immediate), c = &z
        bytecode[pcs_ar, 0] = M(np.where(kind == 2, ar + 3, ar))
        bytecode[pcs_ar, 1] = M(np.where(kind == 2, 7, ar + 1))
        bytecode[pcs_ar, 2] = M(ar + 2)
        bytecode[pcs_ar, 4] = np.where(kind == 2, ONE, 0)           # flag_b (DEREF: addr_b is not fp + operand_b)
        bytecode[pcs_ar, 8] = np.where(kind == 1, ONE, 0)           # mul
        bytecode[pcs_ar, 10] = M(np.where(kind == 0, 1, np.where(kind == 2, 2, 0)))  # aux
    # ---- optional extension-op calls ------------------------------------------------------------------------------------
    ext_rows, pc = [], n_calls + n_arith
    for op, is_be, size, count in ext_calls:
        la, lb = count * size * (1 if is_be else 5) + 4, count * size * 5
        pa, pb, pr = top, top + la, top + la + lb
        top = pr + 5 * count
        assert top + 8 <= mem_len, "memory too small for the extension-op operands"
        memory[pa:pa + la + lb] = ob.rand_field(rng, la + lb)
        r, (qa, qb, qr, aux) = _ext_rows(orc, M, memory, op, is_be, size, count, pa, pb, pr)
        ext_rows.append(r)
        p = pc + np.arange(count)
        bytecode[p, 0], bytecode[p, 1], bytecode[p, 2] = M(qa), M(qb), M(qr)
        bytecode[p, 3:6] = ONE
        bytecode[p, 11] = M(aux)
        pc += count
    # ---- execution table (24 columns: 20 committed + is_precompile, nu_a, nu_b, nu_c) -----------------------------
    ex = np.zeros((24, n_exec), dtype=np.uint32)
    pcs = np.minimum(np.arange(n_exec), ending_pc)
    ex[0] = M(pcs)
    ex[8:20] = bytecode[pcs, :12].T
    ex[2:5] = int(M(Z))                       # addr_a/b/c -> zero vector, values 0
    ex[20, :n_calls] = ONE                    # is_precompile
    ex[21, :n_calls], ex[22, :n_calls], ex[23, :n_calls] = M(addr_a), M(addr_b), M(addr_r)
    ex[21, n_instr:] = ONE                    # nu_a = 1 (jump condition)
    ex[22, n_instr:] = int(M(ending_pc))      # nu_b = jump destination
    if n_arith:
        s_ = slice(n_calls, n_calls + n_arith)
        a_addr = np.where(kind == 2, ar + 3, ar)
        b_addr = ar + 1                        # DEREF: value_a + operand_b = ar + 1 as well
        ex[2, s_], ex[3, s_], ex[4, s_] = M(a_addr), M(b_addr), M(ar + 2)
        ex[5, s_], ex[6, s_], ex[7, s_] = memory[a_addr], memory[b_addr], memory[ar + 2]
        ex[21, s_] = memory[a_addr]                                    # nu_a = value_a
        ex[22, s_] = np.where(kind == 2, int(M(7)), memory[b_addr])    # nu_b = value_b (DEREF: operand_b, flag_b = 1)
        ex[23, s_] = memory[ar + 2]                                    # nu_c = value_c
    if n_ext_calls:
        s_ = slice(n_calls + n_arith, n_instr)
        ex[20, s_] = ONE
        ex[21, s_], ex[22, s_], ex[23, s_] = ex[8, s_], ex[9, s_], ex[10, s_]  # immediates
    # ---- poseidon table (111 columns: 109 committed + index_input_left, precompile_data) -------------------------
    rows = np.zeros((n_pos, 109), dtype=np.uint32)
    left = np.concatenate([addr_a, np.full(n_pos - n_calls, Z)])
    rows[:, 6] = M(left)
    rows[:, 7] = M(left + 4)
    rows[:n_calls, 0] = ONE
    rows[:, 1] = M(np.concatenate([addr_b, np.full(n_pos - n_calls, Z)]))
    rows[:, 2] = M(np.concatenate([addr_r, np.full(n_pos - n_calls, NULL)]))
    rows[:n_calls, 9:25] = inputs[call_blk]
    rows = np.ascontiguousarray(rows)
    if fill_rows is None:
        import ctypes
        orc.lib.orc_poseidon16_fill_rows(rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n_pos))
    else:
        fill_rows(rows)
    pos = np.zeros((111, n_pos), dtype=np.uint32)
    pos[:109] = rows.T
    pos[109] = M(left)
    pos[110] = ONE
    # ---- extension_op table: padding rows (31 columns) ---------------------------------------------------------------
    ext = np.zeros((31, n_ext), dtype=np.uint32)
    ext[1] = ONE            # start
    ext[2] = ONE            # len
    ext[30] = int(M(64))    # aux = EXT_OP_LEN_MULTIPLIER * len
    ext[6] = ext[7] = ext[13] = int(M(Z))
    if ext_rows:
        r = np.concatenate(ext_rows, axis=1)
        assert r.shape[1] < n_ext, "the ExtensionOp table needs at least one padding row (air.rs:147: len - len_shift - 1)"
        ext[:, :r.shape[1]] = r
    # ---- access counters (prove_execution.rs:91-110) -----------------------------------------------------------------
    tables = {0: ex, 1: ext, 2: pos}
    memory_acc, bytecode_acc = access_counters(orc, tables, mem_len, 1 << log_bytecode)
    return dict(log_inv_rate=1, log_memory=log_memory, log_bytecode=log_bytecode, ending_pc=ending_pc, public_memory_size=n_pub,
                public_input=public_input, bytecode_hash=ob.rand_field(rng, 8), bytecode=np.ascontiguousarray(bytecode),
                bytecode_acc=bytecode_acc, memory=memory, memory_acc=memory_acc, tables=tables,
                log_rows={0: log_exec, 1: log_ext, 2: log_pos})


def access_counters(orc, tables, mem_len, bytecode_len):
    """memory_acc / bytecode_acc of prove_execution.rs:91-110 as field elements."""
    M = lambda x: orc.to_monty(np.asarray(x, dtype=np.uint64))  # noqa: E731
    canon = lambda col: orc.from_monty_fast(col).astype(np.int64)  # noqa: E731
    memory_acc = np.zeros(mem_len, dtype=np.int64)
    for t, cols in tables.items():
        for idx, vals in ob.VM_LOOKUPS[t]:
            addr = canon(cols[idx])
            for j in range(len(vals)):
                memory_acc += np.bincount(addr + j, minlength=mem_len)
    bytecode_acc = np.bincount(canon(tables[0][0]), minlength=bytecode_len).astype(np.int64)
    return M(memory_acc), M(bytecode_acc)


def vm_log(w):
    """(pcs, fps) of the straight-line program: the VM's execution log that get_execution_trace starts from."""
    n = w["tables"][0].shape[1]
    return np.minimum(np.arange(n), w["ending_pc"]).astype(np.uint32), np.zeros(n, dtype=np.uint32)


def with_execution_table(orc, w, ex):
    """the witness with its execution table replaced (and the access counters recomputed)"""
    w2 = dict(w, tables=dict(w["tables"]))
    w2["tables"][0] = ex
    w2["memory_acc"], w2["bytecode_acc"] = access_counters(orc, w2["tables"], w["memory"].size, w["bytecode_acc"].size)
    return w2


def header(w):
    return np.array([w["log_inv_rate"], w["log_memory"], w["log_bytecode"], w["ending_pc"], w["public_memory_size"],
                     w["public_input"].size, w["log_rows"][0], w["log_rows"][1], w["log_rows"][2]], dtype=np.uint32)


# all six ExtensionOp modes with single-row and multi-row calls (add_ee/add_be/dot_product_ee/dot_product_be/poly_eq_ee/
# poly_eq_be, extension_op/mod.rs:60-70)
ALL_EXT_MODES = [("add", False, 1, 3), ("mul", False, 4, 5), ("mul", True, 7, 3), ("poly_eq", False, 6, 4),
                 ("poly_eq", True, 3, 2), ("add", True, 2, 2), ("mul", False, 1, 2)]


def build_mixed(orc, rng, **kw):
    """Poseidon calls + ADD/MUL/DEREF instructions + every ExtensionOp mode: the program shape of the recursion workloads."""
    args = dict(n_calls=40, n_blocks=8, log_exec=9, log_pos=8, log_ext=8, log_memory=16, log_bytecode=9, n_arith=30,
                ext_calls=ALL_EXT_MODES)
    args.update(kw)
    return build(orc, rng, **args)


def stacked_n_vars(w):
    """compute_stacked_n_vars (sub_protocols/src/stacked_pcs.rs:183-196) of a witness dict"""
    lr = w["log_rows"]
    total = (2 << w["log_memory"]) + (1 << max(w["log_bytecode"], max(lr.values())))
    total += (20 << lr[0]) + (29 << lr[1]) + (109 << lr[2])
    return int(total - 1).bit_length()
