"""leanVM host side: an assembler for bytecode (what the reference's zkDSL compiler emits: instructions + hints), the
`Bytecode` / witness containers and the binding of the runner (include/leanmultisig_host.h, "leanVM" section).

The reference compiles its programs from a Python-like DSL (crates/lean_compiler, out of scope); programs here are written
at the ISA level (crates/lean_vm/src/isa/instruction.rs:18-58) with this assembler.  `Program.finalize()` produces exactly
what `compile_to_low_level_bytecode` does at its end (lean_compiler/src/c_compile_final.rs:100-180): the code padded with
unreachable instructions to a power of two, the self-jump at `ending_pc = size - 1`, and `instructions_multilinear` =
field_representation of every instruction (lean_compiler/src/instruction_encoder.rs:4-113) in rows of 16 words.

Operands:  K(c) constant (canonical integer, or a label),  M(off) = m[fp + off],  FP(off) = fp + off.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import P, LmError

ARG_CONST, ARG_MEM, ARG_FP = 0, 1, 2
(HINT_INVERSE, HINT_REQUEST_MEMORY, HINT_DEREF, HINT_DECOMPOSE_BITS_XMSS, HINT_DECOMPOSE_BITS_MERKLE_WHIR, HINT_DECOMPOSE_BITS,
 HINT_LESS_THAN, HINT_LOG2_CEIL, HINT_WITNESS_INLINE, HINT_WITNESS_INDIRECT, HINT_PARALLEL_BATCH_START, HINT_DEBUG_ASSERT) = range(1, 13)

# precompile_data (lean_vm/src/tables/poseidon_16/mod.rs:94-98, extension_op/mod.rs:10-14)
POSEIDON_PRECOMPILE_DATA, POSEIDON_PERMUTE, POSEIDON_HALF_OUTPUT, POSEIDON_HARDCODED_LEFT, POSEIDON_OFFSET = 1, 2, 4, 8, 16
EXT_IS_BE, EXT_ADD, EXT_MUL, EXT_POLY_EQ, EXT_LEN = 4, 8, 16, 32, 64
EXT_OPS = {"add": EXT_ADD, "dot_product": EXT_MUL, "mul": EXT_MUL, "poly_eq": EXT_POLY_EQ}

MIN_BYTECODE_LOG_SIZE = 8


def to_monty(x):
    """canonical -> Montgomery (R = 2^32), vectorised"""
    return ((np.asarray(x, dtype=np.uint64) % np.uint64(P)) << np.uint64(32)) % np.uint64(P)


def from_monty(x):
    """Montgomery -> canonical: x * 2^-32 mod p"""
    rinv = pow(1 << 32, P - 2, P)
    return (np.asarray(x, dtype=np.uint64) * np.uint64(rinv)) % np.uint64(P)  # x < 2^32, rinv < 2^31


class Label:
    """a code address known at finalize(); `Label('x') + 3` is an address too (jump tables)"""

    def __init__(self, name, offset=0, scale=1):
        self.name, self.offset = name, offset

    def __add__(self, k):
        return Label(self.name, self.offset + int(k))


class Operand:
    __slots__ = ("mode", "value")

    def __init__(self, mode, value):
        self.mode, self.value = mode, value

    def __repr__(self):
        return {ARG_CONST: "K", ARG_MEM: "M", ARG_FP: "FP"}[self.mode] + f"({self.value})"


def K(c):
    return Operand(ARG_CONST, c)


def M(off):
    assert off >= 0
    return Operand(ARG_MEM, int(off))


def FP(off):
    assert off >= 0
    return Operand(ARG_FP, int(off))


class VmHint(C.Structure):
    """lm_vm_hint"""
    _fields_ = [("pc", C.c_uint32), ("kind", C.c_uint32), ("args", C.c_uint32 * 4), ("mode", C.c_uint8 * 4)]


class VmWitness(C.Structure):
    """lm_vm_witness"""
    _fields_ = [("preamble_memory_len", C.c_uint32), ("n_names", C.c_uint32), ("name_entry_begin", C.c_void_p),
                ("entry_offset", C.c_void_p), ("data", C.c_void_p)]


class VmExecutionView(C.Structure):
    """lm_vm_execution_view"""
    _fields_ = [("n_cycles", C.c_uint64), ("pcs", C.c_void_p), ("fps", C.c_void_p), ("memory_len", C.c_uint64), ("memory", C.c_void_p),
                ("memory_defined", C.c_void_p), ("public_memory_size", C.c_uint64), ("runtime_memory_size", C.c_uint64),
                ("n_poseidon_calls", C.c_uint64), ("poseidon_calls", C.c_void_p), ("n_extension_rows", C.c_uint64),
                ("extension_rows", C.c_void_p), ("n_add", C.c_uint64), ("n_mul", C.c_uint64), ("n_deref", C.c_uint64),
                ("n_jump", C.c_uint64)]


POSEIDON_CALL_WORDS, EXTENSION_ROW_WORDS = 9, 24


class Program:
    """ISA-level assembler.  Instructions are appended in program order; hints attach to the NEXT instruction."""

    def __init__(self):
        self.rows = []     # per instruction: [op_a, op_b, op_c, flag_a, flag_b, flag_c, flag_c_fp, flag_ab_fp, mul, jump, aux, pdata]
        self.fix = []      # (pc, column, Label)
        self.hints = []    # (pc, kind, args, modes)
        self.hint_fix = [] # (hint index, argument, Label)
        self.labels = {}
        self.names = {}    # hint-witness name -> id
        self.starting_frame_memory = 0

    # ---- layout -----------------------------------------------------------------------------------------------------------
    def here(self):
        return len(self.rows)

    def label(self, name):
        assert name not in self.labels, name
        self.labels[name] = self.here()
        return Label(name)

    def name_id(self, name):
        return self.names.setdefault(name, len(self.names))

    # ---- operand encoding (set_nu_a / set_nu_b / set_nu_c of instruction_encoder.rs) -----------------------------------------
    def _ab(self, row, col, flag_col, op, pc):
        assert op.mode in (ARG_CONST, ARG_MEM), f"operand {op} must be a constant or m[fp + x]"
        if op.mode == ARG_CONST:
            row[flag_col] = 1
        self._val(row, col, op, pc)

    def _c(self, row, op, pc):
        if op.mode == ARG_CONST:
            row[5] = 1
        elif op.mode == ARG_FP:
            row[6] = 1
        self._val(row, 2, op, pc)

    def _val(self, row, col, op, pc):
        if isinstance(op.value, Label):
            self.fix.append((pc, col, op.value))
            row[col] = 0
        else:
            row[col] = int(op.value) % P

    def _emit(self, row):
        self.rows.append(row)
        return len(self.rows) - 1

    # ---- instructions ---------------------------------------------------------------------------------------------------------
    def _computation(self, is_mul, a, c, res):
        row, pc = [0] * 12, self.here()
        if is_mul:
            row[8] = 1
        else:
            row[10] = 1
        self._ab(row, 0, 3, a, pc)
        self._ab(row, 1, 4, res, pc)
        self._c(row, c, pc)
        return self._emit(row)

    def add(self, a, c, res):
        """res = a + c   (Computation { Add, arg_a: a, arg_c: c, res }); any ONE unknown operand is solved for"""
        return self._computation(False, a, c, res)

    def mul(self, a, c, res):
        return self._computation(True, a, c, res)

    def deref(self, shift_0, shift_1, res):
        """res = m[m[fp + shift_0] + shift_1]  (or the store m[m[fp + shift_0] + shift_1] = res when res is known)"""
        row, pc = [0] * 12, self.here()
        row[10] = 2
        row[0], row[4], row[1] = int(shift_0), 1, int(shift_1)
        self._c(row, res, pc)
        return self._emit(row)

    def jump(self, cond, dest, new_fp):
        """if cond != 0: pc = dest, fp = new_fp"""
        row, pc = [0] * 12, self.here()
        row[9] = 1
        self._ab(row, 0, 3, cond, pc)
        self._ab(row, 1, 4, dest, pc)
        self._c(row, new_fp, pc)
        return self._emit(row)

    def _precompile(self, pdata, a, b, res):
        row, pc = [0] * 12, self.here()
        row[11] = pdata
        if a.mode == ARG_FP or b.mode == ARG_FP:
            assert a.mode == ARG_FP and b.mode == ARG_FP, "precompile operands a, b: both fp-relative or neither (flag_ab_fp)"
            row[7] = 1
            row[0], row[1] = a.value, b.value
        else:
            self._ab(row, 0, 3, a, pc)
            self._ab(row, 1, 4, b, pc)
        self._c(row, res, pc)
        return self._emit(row)

    def poseidon16(self, a, b, res, half=False, left=None, permute=False):
        """poseidon16_compress variants (PrecompileCompTimeArgs::Poseidon16): operands are ADDRESSES"""
        assert not (permute and (half or left is not None))
        pd = POSEIDON_PRECOMPILE_DATA + POSEIDON_PERMUTE * permute + POSEIDON_HALF_OUTPUT * half
        if left is not None:
            pd += POSEIDON_HARDCODED_LEFT + POSEIDON_OFFSET * int(left)
        return self._precompile(pd, a, b, res)

    def extension_op(self, op, a, b, res, size=1, is_be=False):
        """add_ee/be, dot_product_ee/be, poly_eq_ee/be over `size` elements"""
        assert size >= 1
        return self._precompile(EXT_OPS[op] + EXT_IS_BE * is_be + EXT_LEN * size, a, b, res)

    def panic(self):
        """IntermediateInstruction::Panic: 0 x fp = 1 (c_compile_final.rs:269-277)"""
        return self.mul(K(0), FP(0), K(1))

    def return_from_main(self, zero_cell):
        """`return` of main (b_compile_intermediate.rs:567-584): pc -> ending_pc, fp -> 0 — the rows behind the last cycle are the
        execution table's padding row, whose fp is 0, and the jump constraint ties fp_next to the jump's updated_fp"""
        self.add(K(0), K(0), M(zero_cell))
        return self.jump(K(1), K(Label("@end_program")), M(zero_cell))

    # ---- hints (attach to the next instruction) ------------------------------------------------------------------------------------
    def _hint(self, kind, ops):
        args, modes = [0] * 4, [0] * 4
        for i, o in enumerate(ops):
            if isinstance(o, Operand):
                modes[i] = o.mode
                o = o.value
            if isinstance(o, Label):
                self.hint_fix.append((len(self.hints), i, o))
            else:
                args[i] = int(o)
        self.hints.append((self.here(), kind, args, modes))

    def hint_inverse(self, arg, res_offset):
        self._hint(HINT_INVERSE, [arg, res_offset])

    def hint_request_memory(self, offset, size):
        self._hint(HINT_REQUEST_MEMORY, [offset, size])

    def hint_deref(self, offset_src, offset_target):
        self._hint(HINT_DEREF, [offset_src, offset_target])

    def hint_decompose_bits_xmss(self, decomposed_ptr, to_decompose_ptr, num, chunk_size):
        self._hint(HINT_DECOMPOSE_BITS_XMSS, [decomposed_ptr, to_decompose_ptr, num, chunk_size])

    def hint_decompose_bits_merkle_whir(self, decomposed_ptr, value, chunk_size):
        self._hint(HINT_DECOMPOSE_BITS_MERKLE_WHIR, [decomposed_ptr, value, chunk_size])

    def hint_decompose_bits(self, to_decompose, memory_index, num_bits):
        self._hint(HINT_DECOMPOSE_BITS, [to_decompose, memory_index, num_bits])

    def hint_less_than(self, a, b, res):
        self._hint(HINT_LESS_THAN, [a, b, res])

    def hint_log2_ceil(self, n, res):
        self._hint(HINT_LOG2_CEIL, [n, res])

    def hint_witness(self, name, offset, indirect=False):
        self._hint(HINT_WITNESS_INDIRECT if indirect else HINT_WITNESS_INLINE, [self.name_id(name), offset])

    def hint_parallel_batch_start(self, n_args, end_value):
        self._hint(HINT_PARALLEL_BATCH_START, [n_args, end_value])

    def hint_debug_assert(self, left, right, kind, preceds_runtime_inequality=False):
        self._hint(HINT_DEBUG_ASSERT, [left, right, {"==": 0, "!=": 1, "<": 2, "<=": 3}[kind], int(preceds_runtime_inequality)])

    # ---- the reference's lowering of a range check `val <= bound` (b_compile_intermediate.rs:673-752): 3 cycles --------------------
    def range_check(self, val_off, bound, aux):
        """m[fp + val_off] <= bound, bound a K(..) or M(..) operand; aux = offset of 3 free cells"""
        self.hint_deref(val_off, aux)
        self.deref(val_off, 0, M(aux))
        self.add(M(val_off), M(aux + 1), bound)
        self.hint_deref(aux + 1, aux + 2)
        self.deref(aux + 1, 0, M(aux + 2))

    # ---- finalize -----------------------------------------------------------------------------------------------------------------------
    def finalize(self, log_size=None):
        n_real = len(self.rows)
        size = 1 << max(MIN_BYTECODE_LOG_SIZE, int(np.ceil(np.log2(n_real + 1))))
        if log_size is not None:
            assert (1 << log_size) >= size, "log_size too small for the program"
            size = 1 << log_size
        ending_pc = size - 1
        assert all(h[0] < n_real for h in self.hints), "a hint is attached behind the last instruction"
        for _ in range(ending_pc - n_real):
            self.panic()
        self.labels["@end_program"] = ending_pc
        self.jump(K(1), K(ending_pc), FP(0))  # Label::EndProgram: jump to itself (c_compile_final.rs:43-49)
        rows = np.array(self.rows, dtype=np.uint64)
        for pc, col, lab in self.fix:
            rows[pc, col] = self.labels[lab.name] + lab.offset
        ml = np.zeros((size, 16), dtype=np.uint32)
        ml[:, :12] = to_monty(rows).astype(np.uint32)
        for hi, ai, lab in self.hint_fix:
            self.hints[hi][2][ai] = self.labels[lab.name] + lab.offset
        hints = sorted(self.hints, key=lambda h: h[0])  # stable: hints of one pc keep their order
        return Bytecode(ml, ending_pc, self.starting_frame_memory, hints, dict(self.names), dict(self.labels))


class Bytecode:
    """`Bytecode` of the reference (lean_vm/src/isa/bytecode.rs:17-31) — instructions_multilinear + hints — and its lmh_bytecode."""

    def __init__(self, multilinear, ending_pc, starting_frame_memory, hints, names, labels=None):
        self.multilinear = np.ascontiguousarray(multilinear, dtype=np.uint32)
        self.size = self.multilinear.shape[0]
        self.log_size = int(np.log2(self.size))
        self.ending_pc, self.starting_frame_memory = int(ending_pc), int(starting_frame_memory)
        self.hints, self.names, self.labels = hints, names, labels or {}
        self.lib = None
        self.h = None
        self._hash = None

    def hint_array(self):
        arr = (VmHint * max(1, len(self.hints)))()
        for i, (pc, kind, args, modes) in enumerate(self.hints):
            arr[i].pc, arr[i].kind = pc, kind
            for k in range(4):
                arr[i].args[k], arr[i].mode[k] = args[k], modes[k]
        return arr

    def handle(self):
        if self.h is None:
            self.lib = capi.load()
            arr = self.hint_array()
            self.h = self.lib.lmh_bytecode_new(self.multilinear.ctypes.data, self.log_size, self.size, self.ending_pc,
                                               self.starting_frame_memory, C.cast(arr, C.c_void_p), len(self.hints), len(self.names))
            if not self.h:
                raise LmError("lmh_bytecode_new: " + self.lib.lm_last_error().decode())
            if self.names:  # the keys of ExecutionWitness::hints by id: witness builders that work by name (lmh_aggregate_type_1_witness)
                by_id = sorted(self.names, key=self.names.get)
                arr = (C.c_char_p * len(by_id))(*[n.encode() for n in by_id])
                if self.lib.lmh_bytecode_set_hint_names(self.h, arr, len(by_id)) != 0:
                    raise LmError(self.lib.lm_last_error().decode())
        return self.h

    def hash(self):
        if self._hash is None:
            out = np.empty(8, dtype=np.uint32)
            capi.load().lmh_bytecode_hash(self.handle(), out.ctypes.data)
            self._hash = out
        return self._hash

    def close(self):
        if self.h:
            self.lib.lmh_bytecode_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Witness:
    """ExecutionWitness (lean_vm/src/execution/runner.rs:17-25): preamble_memory_len + named hint streams (Montgomery words)."""

    def __init__(self, bytecode, preamble_memory_len, hints):
        """hints: {name: [array, ...]}; every name of the bytecode must be present (possibly with no entries)"""
        self.preamble_memory_len = int(preamble_memory_len)
        n = len(bytecode.names)
        by_id = [None] * n
        for name, i in bytecode.names.items():
            by_id[i] = [np.ascontiguousarray(e, dtype=np.uint32).reshape(-1) for e in hints.get(name, [])]
        unknown = set(hints) - set(bytecode.names)
        assert all(len(hints[u]) == 0 for u in unknown), f"hint streams the program never reads: {sorted(unknown)}"
        self.name_entry_begin = np.zeros(n + 1, dtype=np.uint64)
        sizes = []
        for i in range(n):
            self.name_entry_begin[i + 1] = self.name_entry_begin[i] + len(by_id[i])
            sizes += [e.size for e in by_id[i]]
        self.entry_offset = np.zeros(len(sizes) + 1, dtype=np.uint64)
        self.entry_offset[1:] = np.cumsum(np.asarray(sizes, dtype=np.uint64)) if sizes else 0
        flat = [e for es in by_id for e in es]
        self.data = np.concatenate(flat) if flat else np.zeros(1, dtype=np.uint32)
        self.c = VmWitness(self.preamble_memory_len, n, self.name_entry_begin.ctypes.data, self.entry_offset.ctypes.data, self.data.ctypes.data)


def _view(ptr, n, dtype=np.uint32):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8 if dtype == np.uint8 else C.c_uint32)), shape=(int(n),)).copy()


class Execution:
    """lmh_execution: the ExecutionResult of one run (copied out on demand)."""

    def __init__(self, lib, handle, lazy=False):
        self.lib, self.h = lib, handle
        self.on_device = bool(lib.lmh_execution_on_device(handle))  # the batches ran on the device: the log is resident in HBM
        self.view = None
        if not lazy:
            self.load()

    def load(self):
        v = VmExecutionView()
        self.lib.lmh_execution_view(self.h, C.byref(v))
        self.view = v
        self.n_cycles, self.memory_len = int(v.n_cycles), int(v.memory_len)
        self.public_memory_size, self.runtime_memory_size = int(v.public_memory_size), int(v.runtime_memory_size)
        self.n_poseidon_calls, self.n_extension_rows = int(v.n_poseidon_calls), int(v.n_extension_rows)
        self.counts = dict(add=int(v.n_add), mul=int(v.n_mul), deref=int(v.n_deref), jump=int(v.n_jump))

    def pcs(self):
        return _view(self.view.pcs, self.n_cycles)

    def fps(self):
        return _view(self.view.fps, self.n_cycles)

    def memory(self):
        return _view(self.view.memory, self.memory_len)

    def memory_defined(self):
        return _view(self.view.memory_defined, self.memory_len, np.uint8)

    def poseidon_calls(self):
        return _view(self.view.poseidon_calls, self.n_poseidon_calls * POSEIDON_CALL_WORDS).reshape(-1, POSEIDON_CALL_WORDS)

    def extension_rows(self):
        return _view(self.view.extension_rows, self.n_extension_rows * EXTENSION_ROW_WORDS).reshape(-1, EXTENSION_ROW_WORDS)

    def close(self):
        if self.h:
            self.lib.lmh_execution_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def execute(bytecode, public_input, witness, n_threads=0, ctx=None, lazy=False):
    """execute_bytecode (runner.rs:57-68): raises LmError with the RunnerError on failure.  ctx: the parallel loop batches run on that
    context's device (lmh_execute_bytecode_device); lazy: do not download the log (Execution.load() does it)."""
    lib = capi.load()
    pi = np.ascontiguousarray(public_input, dtype=np.uint32)
    out = C.c_void_p()
    if ctx is None:
        rc = lib.lmh_execute_bytecode(bytecode.handle(), pi.ctypes.data, pi.size, C.byref(witness.c), n_threads, C.byref(out))
    else:
        rc = lib.lmh_execute_bytecode_device(ctx.h, bytecode.handle(), pi.ctypes.data, pi.size, C.byref(witness.c), n_threads, C.byref(out))
    if rc != 0:
        raise LmError(lib.lm_last_error().decode())
    return Execution(lib, out.value, lazy=lazy)


def poseidon16_compress_many(states, n_threads=0):
    """(n, 16) Montgomery words -> perm(x) + x on the host thread pool (lmh_poseidon16_compress_many)"""
    s = np.ascontiguousarray(states, dtype=np.uint32).reshape(-1, 16).copy()
    capi.load().lmh_poseidon16_compress_many(s.ctypes.data, s.shape[0], n_threads)
    return s


class DeviceTrace:
    """lmh_vm_trace: get_execution_trace on the device (every table column, the padded memory image) from an Execution."""

    def __init__(self, ctx, bytecode, execution, public_input, log_inv_rate=1):
        self.ctx, self.lib = ctx, ctx.lib
        pi = np.ascontiguousarray(public_input, dtype=np.uint32)
        out = C.c_void_p()
        rc = self.lib.lmh_get_execution_trace(ctx.h, bytecode.handle(), execution.h, pi.ctypes.data, pi.size, log_inv_rate, C.byref(out))
        if rc != 0:
            raise LmError(self.lib.lm_last_error().decode())
        self.h = out.value
        self.bytecode = bytecode  # owns the device copy of the bytecode table
        self.view = C.cast(self.lib.lmh_vm_trace_view(self.h), C.POINTER(capi.ExecutionTrace)).contents

    def table(self, t):
        """table t as (n_columns_total, 2^log_rows) host array"""
        n_total = {0: 24, 1: 31, 2: 111}[t]
        tb = self.view.tables[t]
        ptrs = np.ctypeslib.as_array(C.cast(tb.d_cols, C.POINTER(C.c_uint64)), shape=(n_total,))
        out = np.empty((n_total, 1 << tb.log_rows), dtype=np.uint32)
        for c in range(n_total):
            self.ctx._check(self.lib.lm_download(self.ctx.h, out[c].ctypes.data, int(ptrs[c]), out.shape[1]))
        return out

    def memory(self):
        out = np.empty(1 << self.view.log_memory, dtype=np.uint32)
        self.ctx._check(self.lib.lm_download(self.ctx.h, out.ctypes.data, self.view.d_memory, out.size))
        return out

    def close(self):
        if self.h and self.ctx.h:
            self.lib.lmh_vm_trace_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VmRunInfo(C.Structure):
    """lm_vm_run_info"""
    _fields_ = [("on_device", C.c_uint32), ("n_device_batches", C.c_uint32), ("n_host_batches", C.c_uint32), ("run_repeated", C.c_uint32),
                ("host_batch_reason", C.c_char * 256)]

    def to_dict(self):
        return dict(vm_on_device=bool(self.on_device) and self.n_host_batches == 0 and not self.run_repeated, device_batches=int(self.n_device_batches),
                    host_batches=int(self.n_host_batches), run_repeated=bool(self.run_repeated), fallback_reason=self.host_batch_reason.decode() or None)


XMSS_SIG_WORDS = 4 + 4 + 6 + 42 * 4 + 32 * 4


def pack_xmss_signatures(sig):
    """{root (n,4), pp (n,4), randomness (n,6), chain_tips (n,42,4), merkle_proof (n,32,4)} (leanmultisig_amd/xmss.py) -> the (n, LM_XMSS_SIG_WORDS)
    array of (XmssPublicKey, XmssSignature) pairs lmh_aggregate_type_1 takes"""
    n = sig["root"].shape[0]
    return np.ascontiguousarray(np.concatenate([sig["root"], sig["pp"], sig["randomness"], sig["chain_tips"].reshape(n, -1),
                                                sig["merkle_proof"].reshape(n, -1)], axis=1), dtype=np.uint32)


class Type1Witness:
    """lmh_type1_witness: what aggregate_type_1 (rec_aggregation/src/type_1_aggregation.rs:206-377) builds before prove_execution"""

    def __init__(self, bytecode, raw_xmss, message, slot):
        self.lib = capi.load()
        raw = np.ascontiguousarray(raw_xmss, dtype=np.uint32).reshape(-1, XMSS_SIG_WORDS)
        msg = np.ascontiguousarray(message, dtype=np.uint32)
        out = C.c_void_p()
        if self.lib.lmh_aggregate_type_1_witness(bytecode.handle(), raw.ctypes.data, raw.shape[0], msg.ctypes.data, int(slot), C.byref(out)) != 0:
            raise LmError(self.lib.lm_last_error().decode())
        self.h = out.value
        self.c = C.cast(self.lib.lmh_type1_witness_vm(self.h), C.POINTER(VmWitness)).contents
        self.public_input = _view(self.lib.lmh_type1_witness_public_input(self.h), 8)
        self.n_sigs = int(self.lib.lmh_type1_witness_n_sigs(self.h))
        nw = C.c_uint64()
        ptr = self.lib.lmh_type1_witness_input_data(self.h, C.byref(nw))
        self.input_data = _view(ptr, nw.value)
        self.pubkeys = _view(self.lib.lmh_type1_witness_pubkeys(self.h), 8 * self.n_sigs).reshape(-1, 8)

    def streams(self):
        """(name_entry_begin, entry_offset, data) as arrays"""
        n = int(self.c.n_names)
        neb = np.ctypeslib.as_array(C.cast(self.c.name_entry_begin, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        eo = np.ctypeslib.as_array(C.cast(self.c.entry_offset, C.POINTER(C.c_uint64)), shape=(int(neb[-1]) + 1,)).copy()
        return neb, eo, _view(self.c.data, int(eo[-1]))

    def close(self):
        if self.h:
            self.lib.lmh_type1_witness_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def aggregate_type_1(ctx, prover, bytecode, raw_xmss, message, slot, builder, n_threads=0):
    """aggregate_type_1(&[], raw_xmss, message, slot, log_inv_rate): input assembly + prove_execution, the proof left in `prover`.
    Returns ([inputs_ms, vm_ms, trace_ms, prove_ms], VmRunInfo)."""
    raw = np.ascontiguousarray(raw_xmss, dtype=np.uint32).reshape(-1, XMSS_SIG_WORDS)
    msg = np.ascontiguousarray(message, dtype=np.uint32)
    times = (C.c_double * 4)()
    info = VmRunInfo()
    rc = ctx.lib.lmh_aggregate_type_1(ctx.h, prover.h, bytecode.handle(), raw.ctypes.data, raw.shape[0], msg.ctypes.data, int(slot), C.byref(builder),
                                      n_threads, times, C.byref(info))
    if rc != 0:
        raise LmError(ctx.lib.lm_last_error().decode())
    return list(times), info


def prove_execution_vm(ctx, prover, bytecode, public_input, witness, builder, n_threads=0):
    """prove_execution(bytecode, public_input, witness, whir_config) (lean_prover/src/prove_execution.rs:20-274), whole: the VM run,
    the trace on the device, the proof (left in `prover`).  Returns [vm_ms, trace_ms, prove_ms]."""
    pi = np.ascontiguousarray(public_input, dtype=np.uint32)
    times = (C.c_double * 3)()
    rc = ctx.lib.lmh_prove_execution_vm(ctx.h, prover.h, bytecode.handle(), pi.ctypes.data, pi.size, C.byref(witness.c), C.byref(builder),
                                        n_threads, times)
    if rc != 0:
        raise LmError(ctx.lib.lm_last_error().decode())
    return list(times)
