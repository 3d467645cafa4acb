"""Synthetic but CONSISTENT execution witness for the proving slice after witness generation (TEST INFRASTRUCTURE).

The reference obtains (memory, tables) by running the zkDSL aggregation program in its VM; that interpreter is out of
scope.  This module builds by hand a tiny straight-line leanVM program that the three AIRs, the memory /
bytecode lookups and the precompile bus all accept, so that prove_execution's slice can be proven AND verified:

  pc = 0 .. N-1 : `poseidon16_compress(a_i, b_i) -> res_i` precompile calls with immediate operands
                  (execution/air.rs:56-129: flag_a = flag_b = flag_c = 1, aux = mul = jump = 0  =>  is_precompile = 1,
                   nu_a/b/c = operand_a/b/c, next pc = pc + 1, fp unchanged)
  pc = N        : the self-loop jump used as padding row (execution/mod.rs:59-74)
Poseidon16 table: one active row per call + padding rows (poseidon_16/mod.rs:176-199); extension_op table: padding
rows only (extension_op/mod.rs:125-135).  Memory holds the inputs, outputs, the zero vector and the null hash.
"""
import numpy as np

from tests import oracle_binding as ob
from tests.oracle_binding import P

ONE = 0x01FFFFFE


def build(orc, rng, n_calls, n_blocks=8, log_exec=8, log_pos=8, log_ext=8, log_memory=16, log_bytecode=8, fill_rows=None):
    """fill_rows(rows) may replace the oracle's Poseidon trace generator for big tables (rows: (n, 109) uint32, in place)."""
    M = lambda x: orc.to_monty(np.asarray(x, dtype=np.uint64))  # noqa: E731
    n_exec, n_pos, n_ext = 1 << log_exec, 1 << log_pos, 1 << log_ext
    assert n_calls < n_exec and n_calls <= n_pos and n_calls < (1 << log_bytecode)
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
    ending_pc = n_calls
    # ---- bytecode (row-major, stride 16; columns = the 12 instruction columns of the execution table) ----------
    # operand_a, operand_b, operand_c, flag_a, flag_b, flag_c, flag_c_fp, flag_ab_fp, mul, jump, aux, precompile_data
    bytecode = np.zeros((1 << log_bytecode, 16), dtype=np.uint32)
    bytecode[:n_calls, 0], bytecode[:n_calls, 1], bytecode[:n_calls, 2] = M(addr_a), M(addr_b), M(addr_r)
    bytecode[:n_calls, 3:6] = ONE
    bytecode[:n_calls, 11] = ONE
    bytecode[ending_pc, :12] = [ONE, int(M(ending_pc)), 0, ONE, ONE, 0, ONE, 0, 0, ONE, 0, 0]
    # ---- execution table (24 columns: 20 committed + is_precompile, nu_a, nu_b, nu_c) -----------------------------
    ex = np.zeros((24, n_exec), dtype=np.uint32)
    pcs = np.minimum(np.arange(n_exec), ending_pc)
    ex[0] = M(pcs)
    ex[8:20] = bytecode[pcs, :12].T
    ex[2:5] = int(M(Z))                       # addr_a/b/c -> zero vector, values 0
    ex[20, :n_calls] = ONE                    # is_precompile
    ex[21, :n_calls], ex[22, :n_calls], ex[23, :n_calls] = M(addr_a), M(addr_b), M(addr_r)
    ex[21, n_calls:] = ONE                    # nu_a = 1 (jump condition)
    ex[22, n_calls:] = int(M(ending_pc))      # nu_b = jump destination
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
    # ---- access counters (prove_execution.rs:91-110) -----------------------------------------------------------------
    tables = {0: ex, 1: ext, 2: pos}
    memory_acc = np.zeros(mem_len, dtype=np.int64)
    canon = lambda col: orc.from_monty_fast(col).astype(np.int64)  # noqa: E731
    for t, cols in tables.items():
        for idx, vals in ob.VM_LOOKUPS[t]:
            addr = canon(cols[idx])
            for j in range(len(vals)):
                memory_acc += np.bincount(addr + j, minlength=mem_len)
    bytecode_acc = np.bincount(canon(ex[0]), minlength=1 << log_bytecode).astype(np.int64)
    return dict(log_inv_rate=1, log_memory=log_memory, log_bytecode=log_bytecode, ending_pc=ending_pc, public_memory_size=n_pub,
                public_input=public_input, bytecode_hash=ob.rand_field(rng, 8), bytecode=np.ascontiguousarray(bytecode),
                bytecode_acc=M(bytecode_acc), memory=memory, memory_acc=M(memory_acc), tables=tables,
                log_rows={0: log_exec, 1: log_ext, 2: log_pos})


def header(w):
    return np.array([w["log_inv_rate"], w["log_memory"], w["log_bytecode"], w["ending_pc"], w["public_memory_size"],
                     w["public_input"].size, w["log_rows"][0], w["log_rows"][1], w["log_rows"][2]], dtype=np.uint32)
