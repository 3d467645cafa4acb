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


def build(orc, rng, n_calls, n_blocks=8, log_exec=8, log_pos=8, log_ext=8, log_memory=16, log_bytecode=8):
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
    blocks = []
    for b in range(n_blocks):
        a0 = base + 32 * b
        inp = ob.rand_field(rng, 16)
        memory[a0:a0 + 16] = inp
        out = orc.poseidon16_compress(inp)[0][:8]
        memory[a0 + 16:a0 + 24] = out      # res: outputs_left; res+8..16 stay 0 (outputs_right = 0 in compress mode)
        blocks.append((a0, a0 + 8, a0 + 16))
    ending_pc = n_calls
    # ---- bytecode (row-major, stride 16; columns = the 12 instruction columns of the execution table) ----------
    bytecode = np.zeros((1 << log_bytecode, 16), dtype=np.uint32)
    calls = [blocks[i % n_blocks] for i in range(n_calls)]
    for pc, (a, b, r) in enumerate(calls):
        # operand_a, operand_b, operand_c, flag_a, flag_b, flag_c, flag_c_fp, flag_ab_fp, mul, jump, aux, precompile_data
        bytecode[pc, :12] = [int(M(a)), int(M(b)), int(M(r)), ONE, ONE, ONE, 0, 0, 0, 0, 0, ONE]
    bytecode[ending_pc, :12] = [ONE, int(M(ending_pc)), 0, ONE, ONE, 0, ONE, 0, 0, ONE, 0, 0]
    # ---- execution table (24 columns: 20 committed + is_precompile, nu_a, nu_b, nu_c) -----------------------------
    ex = np.zeros((24, n_exec), dtype=np.uint32)
    for row in range(n_exec):
        pc = min(row, ending_pc)
        ex[0, row] = int(M(pc))
        ex[8:20, row] = bytecode[pc, :12]
        ex[2:5, row] = int(M(Z))                 # addr_a/b/c -> zero vector, values 0
        if pc < n_calls:
            a, b, r = calls[pc]
            ex[20, row] = ONE                     # is_precompile
            ex[21, row], ex[22, row], ex[23, row] = int(M(a)), int(M(b)), int(M(r))
        else:
            ex[21, row] = ONE                     # nu_a = 1 (jump condition)
            ex[22, row] = int(M(ending_pc))       # nu_b = jump destination
    # ---- poseidon table (111 columns: 109 committed + index_input_left, precompile_data) -------------------------
    rows = np.zeros((n_pos, 109), dtype=np.uint32)
    left = np.array([c[0] for c in calls] + [Z] * (n_pos - n_calls))
    rows[:, 6] = M(left)
    rows[:, 7] = M(left + 4)
    rows[:n_calls, 0] = ONE
    rows[:, 1] = M([c[1] for c in calls] + [Z] * (n_pos - n_calls))
    rows[:, 2] = M([c[2] for c in calls] + [NULL] * (n_pos - n_calls))
    for i, (a, b, r) in enumerate(calls):
        rows[i, 9:25] = memory[a:a + 16]
    rows = np.ascontiguousarray(rows)
    import ctypes
    orc.lib.orc_poseidon16_fill_rows(rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n_pos))
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
    canon = lambda col: orc.from_monty(col).astype(np.int64)  # noqa: E731
    for t, cols in tables.items():
        for idx, vals in ob.VM_LOOKUPS[t]:
            addr = canon(cols[idx])
            for j in range(len(vals)):
                np.add.at(memory_acc, addr + j, 1)
    bytecode_acc = np.zeros(1 << log_bytecode, dtype=np.int64)
    np.add.at(bytecode_acc, canon(ex[0]), 1)
    return dict(log_inv_rate=1, log_memory=log_memory, log_bytecode=log_bytecode, ending_pc=ending_pc, public_memory_size=n_pub,
                public_input=public_input, bytecode_hash=ob.rand_field(rng, 8), bytecode=np.ascontiguousarray(bytecode),
                bytecode_acc=M(bytecode_acc), memory=memory, memory_acc=M(memory_acc), tables=tables,
                log_rows={0: log_exec, 1: log_ext, 2: log_pos})


def header(w):
    return np.array([w["log_inv_rate"], w["log_memory"], w["log_bytecode"], w["ending_pc"], w["public_memory_size"],
                     w["public_input"].size, w["log_rows"][0], w["log_rows"][1], w["log_rows"][2]], dtype=np.uint32)
