// TEST INFRASTRUCTURE ONLY — CPU restatement of the leanVM runner and of get_execution_trace, the checker of
// leanmultisig_amd/csrc/host/lm_vm.cpp + lm_node.cpp.  Never linked by the product.
//
// Follows, function by function:
//   crates/lean_vm/src/isa/operands/{mem_or_constant,mem_or_fp_or_constant}.rs   MemOrConstant / MemOrFpOrConstant
//   crates/lean_vm/src/isa/instruction.rs:146-246                                execute_instruction
//   crates/lean_vm/src/isa/hint.rs:137-203,270-386                               CustomHint::execute, Hint::execute_hint
//   crates/lean_vm/src/execution/memory.rs                                       Memory, SegmentMemory
//   crates/lean_vm/src/execution/runner.rs:121-482                               run_loop, resolve_deref_hints,
//                                                                                execute_bytecode_helper, handle_parallel_batch
//   crates/lean_vm/src/tables/poseidon_16/mod.rs:209-289                         Poseidon16Precompile::execute
//   crates/lean_vm/src/tables/extension_op/exec.rs                               exec_multi_row, solve_unknowns, fill_trace_extension_op
//   crates/lean_prover/src/trace_gen.rs                                          get_execution_trace, pad_table
// The parallel segments of a batch are executed one after the other with the reference's SegmentMemory semantics (reads of
// other segments fail, writes outside the segment are deferred) — rayon's scheduling does not change the result.
// Parity: unpinned by reference-held vectors (the reference holds none for the VM); pinned by its own unit behaviours
// (tests/test_vm.py: every instruction / operand-unknown case / hint) and by the AIR + lookups accepting the traces it produces.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "kb_oracle.hpp"
#include "air_oracle.hpp"  // poseidon16_fill_row (generate_trace_rows_for_perm on one 109-word row)

namespace orc {
namespace vm {

using F = uint32_t;  // Montgomery word
static inline size_t to_usize(F x) { return from_monty(x); }
static inline F from_usize(size_t x) { return to_monty((uint32_t)(x % P)); }

struct RunnerError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- operands -------------------------------------------------------------------------------------------------------------------
enum class Arg { Constant = 0, MemoryAfterFp = 1, FpRelative = 2 };
struct Operand {  // MemOrConstant (no FpRelative) / MemOrFpOrConstant
    Arg kind = Arg::Constant;
    F constant = 0;     // Constant(c)
    size_t offset = 0;  // MemoryAfterFp { offset } / FpRelative { offset }
};

// ---- memory ---------------------------------------------------------------------------------------------------------------------
static const size_t MAX_LOG_MEMORY_SIZE = 26, MIN_LOG_MEMORY_SIZE = 16, MIN_LOG_N_ROWS_PER_TABLE = 8;
struct MemoryAccess {
    virtual ~MemoryAccess() {}
    virtual std::optional<F> try_get(size_t index) const = 0;
    virtual void set(size_t index, F value) = 0;
    F get(size_t index) const {
        auto v = try_get(index);
        if (!v) throw RunnerError("UndefinedMemory(" + std::to_string(index) + ")");
        return *v;
    }
    std::vector<F> get_slice(size_t start, size_t len) const {
        std::vector<F> out(len);
        for (size_t i = 0; i < len; i++) out[i] = get(start + i);
        return out;
    }
    void set_slice(size_t start, const F* values, size_t n) {
        for (size_t i = 0; i < n; i++) set(start + i, values[i]);
    }
    EF get_ef_element(size_t index) const {
        EF r;
        for (int k = 0; k < 5; k++) r.v[k] = get(index + k);
        return r;
    }
    std::optional<EF> try_get_ef_element(size_t index) const {
        EF r;
        for (int k = 0; k < 5; k++) {
            auto v = try_get(index + k);
            if (!v) return std::nullopt;
            r.v[k] = *v;
        }
        return r;
    }
    void set_ef_element(size_t index, const EF& v) { set_slice(index, v.v, 5); }
    void make_slices_equal_and_defined(size_t p0, size_t p1, size_t len) {  // memory.rs:41-66
        for (size_t i = 0; i < len; i++) {
            auto a = try_get(p0 + i), b = try_get(p1 + i);
            if (a && b) {
                if (*a != *b) throw RunnerError("NotEqual");
            } else if (a)
                set(p1 + i, *a);
            else if (b)
                set(p0 + i, *b);
            else {
                set(p0 + i, 0);
                set(p1 + i, 0);
            }
        }
    }
};
struct Memory : MemoryAccess {
    std::vector<std::optional<F>> cells;
    std::optional<F> try_get(size_t index) const override { return index < cells.size() ? cells[index] : std::nullopt; }
    void set(size_t index, F value) override {
        if (index >= cells.size()) {
            if (index >= ((size_t)1 << MAX_LOG_MEMORY_SIZE)) throw RunnerError("OutOfMemory");
            cells.resize(index + 1);
        }
        if (cells[index]) {
            if (*cells[index] != value) throw RunnerError("MemoryAlreadySet { address: " + std::to_string(index) + " }");
        } else
            cells[index] = value;
    }
};
struct SegmentMemory : MemoryAccess {  // memory.rs:118-189
    const std::vector<std::optional<F>>* all;  // the main memory; [0, shared_len) is the shared read-only part
    size_t shared_len, segment_start, segment_len;
    std::vector<std::optional<F>> segment;     // private copy of this segment's slice, written back by the caller
    std::vector<std::pair<size_t, F>> deferred_writes;
    std::optional<F> try_get(size_t index) const override {
        if (index < segment_start) return index < shared_len ? (*all)[index] : std::nullopt;
        const size_t off = index - segment_start;
        return off < segment_len ? segment[off] : std::nullopt;
    }
    void set(size_t index, F value) override {
        const bool in_segment = index >= segment_start && index - segment_start < segment_len;
        if (!in_segment) {
            deferred_writes.emplace_back(index, value);
            return;
        }
        auto& c = segment[index - segment_start];
        if (c) {
            if (*c != value) throw RunnerError("MemoryAlreadySet { address: " + std::to_string(index) + " }");
        } else
            c = value;
    }
};

static inline std::optional<F> try_read(const Operand& o, const MemoryAccess& m, size_t fp) {
    switch (o.kind) {
        case Arg::Constant: return o.constant;
        case Arg::MemoryAfterFp: return m.try_get(fp + o.offset);
        default: return from_usize(fp + o.offset);
    }
}
static inline F read_value(const Operand& o, const MemoryAccess& m, size_t fp) {
    auto v = try_read(o, m, fp);
    if (!v) throw RunnerError("UndefinedMemory(" + std::to_string(fp + o.offset) + ")");
    return *v;
}
static inline bool is_value_unknown(const Operand& o, const MemoryAccess& m, size_t fp) { return !try_read(o, m, fp); }
static inline size_t memory_address(const Operand& o, size_t fp) {
    if (o.kind != Arg::MemoryAfterFp) throw RunnerError("NotAPointer");
    return fp + o.offset;
}

// ---- program --------------------------------------------------------------------------------------------------------------------
enum class Op { Add, Mul, Deref, Jump, Poseidon16, ExtensionOp };
struct Instruction {
    Op op;
    Operand a, b, c;  // Computation: arg_a, res, arg_c | Deref: shift_0 (a.offset), shift_1 (b.constant), res (c) |
                      // Jump: condition, dest, updated_fp | Precompile: arg_0, arg_1, res
    bool half_output = false, permute = false, hardcoded = false;  // Poseidon16
    size_t hardcoded_offset_left = 0;
    size_t size = 0;                                               // ExtensionOp
    bool is_be = false;
    int ext_op = 0;  // 8 add, 16 mul, 32 poly_eq
};
struct Hint {
    uint32_t kind;
    uint32_t args[4];
    uint8_t mode[4];
};
struct Bytecode {
    std::vector<Instruction> code;
    std::vector<std::vector<Hint>> hints;  // per pc
    const uint32_t* instructions_multilinear = nullptr;
    size_t log_size = 0, ending_pc = 0, starting_frame_memory = 0, n_names = 0;
};

// inverse of field_representation (lean_compiler/src/instruction_encoder.rs:4-113)
static inline Instruction decode(const uint32_t* row) {
    uint32_t c[12];
    for (int k = 0; k < 12; k++) c[k] = from_monty(row[k]);
    auto ab = [&](int col, uint32_t flag, uint32_t flag_fp) {
        Operand o;
        if (flag) {
            o.kind = Arg::Constant;
            o.constant = row[col];
        } else {
            o.kind = flag_fp ? Arg::FpRelative : Arg::MemoryAfterFp;
            o.offset = c[col];
        }
        return o;
    };
    Instruction in;
    in.a = ab(0, c[3], c[7]);
    in.b = ab(1, c[4], c[7]);
    in.c = ab(2, c[5], c[6]);
    const uint32_t mul = c[8], jump = c[9], aux = c[10], pd = c[11];
    if (pd) {
        if (pd & 1) {
            in.op = Op::Poseidon16;
            in.permute = (pd >> 1) & 1;
            in.half_output = (pd >> 2) & 1;
            in.hardcoded = (pd >> 3) & 1;
            in.hardcoded_offset_left = pd >> 4;
        } else {
            in.op = Op::ExtensionOp;
            in.is_be = (pd >> 2) & 1;
            in.ext_op = pd & 56;
            in.size = pd >> 6;
        }
    } else if (jump)
        in.op = Op::Jump;
    else if (mul)
        in.op = Op::Mul;
    else if (aux == 1)
        in.op = Op::Add;
    else if (aux == 2)
        in.op = Op::Deref;
    else
        throw std::runtime_error("undecodable instruction");
    return in;
}

// ---- tables as the runner fills them ----------------------------------------------------------------------------------------------
struct Trace {
    std::vector<size_t> pcs, fps;
    std::vector<std::vector<F>> poseidon = std::vector<std::vector<F>>(111);  // columns 0..8, 9..24, 109, 110 pushed by execute
    std::vector<std::vector<F>> extension = std::vector<std::vector<F>>(31);
    size_t add = 0, mul = 0, deref = 0, jump = 0;
    std::vector<std::pair<size_t, size_t>> pending_deref_hints;
    void merge(Trace&& o) {
        pcs.insert(pcs.end(), o.pcs.begin(), o.pcs.end());
        fps.insert(fps.end(), o.fps.begin(), o.fps.end());
        add += o.add, mul += o.mul, deref += o.deref, jump += o.jump;
        pending_deref_hints.insert(pending_deref_hints.end(), o.pending_deref_hints.begin(), o.pending_deref_hints.end());
        for (size_t c = 0; c < poseidon.size(); c++) poseidon[c].insert(poseidon[c].end(), o.poseidon[c].begin(), o.poseidon[c].end());
        for (size_t c = 0; c < extension.size(); c++) extension[c].insert(extension[c].end(), o.extension[c].begin(), o.extension[c].end());
    }
};

struct WitnessHints {
    size_t preamble_memory_len = 0;
    const uint64_t* name_entry_begin = nullptr;
    const uint64_t* entry_offset = nullptr;
    const uint32_t* data = nullptr;
};

// ---- precompiles -----------------------------------------------------------------------------------------------------------------------
static inline void poseidon16_execute(const Instruction& in, F arg_a, F arg_b, F index_res_a, MemoryAccess& memory, Trace& tr) {
    if (in.permute && (in.half_output || in.hardcoded)) throw RunnerError("Panic: permute is mutually exclusive with half_output / hardcoded_left");
    const size_t arg_a_usize = to_usize(arg_a);
    const size_t left_first_addr = in.hardcoded ? in.hardcoded_offset_left : arg_a_usize;
    const size_t left_second_addr = in.hardcoded ? arg_a_usize : arg_a_usize + 4;
    auto first = memory.get_slice(left_first_addr, 4), second = memory.get_slice(left_second_addr, 4), right = memory.get_slice(to_usize(arg_b), 8);
    F input[16];
    std::memcpy(input, first.data(), 16);
    std::memcpy(input + 4, second.data(), 16);
    std::memcpy(input + 8, right.data(), 32);
    const size_t res_addr = to_usize(index_res_a);
    F st[16];
    std::memcpy(st, input, 64);
    if (in.permute) {
        poseidon16_permute(st);
        memory.set_slice(res_addr, st, 16);
    } else {
        poseidon16_compress(st);
        memory.set_slice(res_addr, st, in.half_output ? 4 : 8);
    }
    const size_t off = in.hardcoded ? in.hardcoded_offset_left : 0;
    auto& c = tr.poseidon;
    c[0].push_back(ONE);
    c[1].push_back(arg_b);
    c[2].push_back(index_res_a);
    c[3].push_back(in.half_output ? ONE : 0);
    c[4].push_back(in.hardcoded ? ONE : 0);
    c[5].push_back(from_usize(off));
    c[6].push_back(from_usize(left_first_addr));
    c[7].push_back(from_usize(left_second_addr));
    c[8].push_back(in.permute ? ONE : 0);
    for (int i = 0; i < 16; i++) c[9 + i].push_back(input[i]);
    c[109].push_back(arg_a);
    c[110].push_back(from_usize(1 + 2 * in.permute + 4 * in.half_output + 8 * in.hardcoded + 16 * off));
}

static inline EF ext_compute_elem(const EF& a, const EF& b, int op) {
    if (op == 8) return ef_add(a, b);
    if (op == 16) return ef_mul(a, b);
    EF ab = ef_mul(a, b);
    EF r = ef_sub(ef_sub(ef_add(ab, ab), a), b);
    r.v[0] = add(r.v[0], ONE);
    return r;
}
static inline void ext_solve_unknowns(F ptr_a, F ptr_b, F ptr_res, bool is_be, int op, MemoryAccess& memory) {
    const size_t addr_a = to_usize(ptr_a), addr_b = to_usize(ptr_b), addr_res = to_usize(ptr_res);
    std::optional<EF> a;
    if (is_be) {
        auto v = memory.try_get(addr_a);
        if (v) a = ef_from_base(*v);
    } else
        a = memory.try_get_ef_element(addr_a);
    auto b = memory.try_get_ef_element(addr_b), c = memory.try_get_ef_element(addr_res);
    if (op == 16 && !is_be) {
        if (b && ef_eq(*b, ef_one())) return memory.make_slices_equal_and_defined(addr_a, addr_res, 5);
        if (a && ef_eq(*a, ef_one())) return memory.make_slices_equal_and_defined(addr_b, addr_res, 5);
    }
    if (a && b && c) {
        if (!ef_eq(ext_compute_elem(*a, *b, op), *c)) throw RunnerError("InvalidExtensionOp");
    } else if (a && b && !c) {
    } else if (!a && b && c) {
        if (op == 32) throw RunnerError("unreachable");
        const EF x = op == 8 ? ef_sub(*c, *b) : ef_mul(*c, ef_inv(*b));
        if (is_be) {
            if (x.v[1] | x.v[2] | x.v[3] | x.v[4]) throw RunnerError("Panic: solved A not in base field");
            memory.set(addr_a, x.v[0]);
        } else
            memory.set_ef_element(addr_a, x);
    } else if (a && !b && c) {
        if (op == 32) throw RunnerError("unreachable");
        memory.set_ef_element(addr_b, op == 8 ? ef_sub(*c, *a) : ef_mul(*c, ef_inv(*a)));
    } else
        throw RunnerError("InvalidExtensionOp");
}
static inline void ext_exec_multi_row(const Instruction& in, F ptr_a, F ptr_b, F ptr_res, MemoryAccess& memory, Trace& tr) {
    const size_t size = in.size;
    const int op = in.ext_op;
    if (size < 1) throw RunnerError("Panic: size >= 1");
    if (size == 1 && op != 32) ext_solve_unknowns(ptr_a, ptr_b, ptr_res, in.is_be, op, memory);
    const size_t a_stride = in.is_be ? 1 : 5;
    std::vector<EF> elems, v_bs, computations(size);
    std::vector<F> idx_as, idx_bs;
    for (size_t i = 0; i < size; i++) {
        const size_t addr_a = to_usize(ptr_a) + i * a_stride, addr_b = to_usize(ptr_b) + i * 5;
        const EF v_a = in.is_be ? ef_from_base(memory.get(addr_a)) : memory.get_ef_element(addr_a);
        const EF v_b = memory.get_ef_element(addr_b);
        elems.push_back(ext_compute_elem(v_a, v_b, op));
        v_bs.push_back(v_b);
        idx_as.push_back(from_usize(addr_a));
        idx_bs.push_back(from_usize(addr_b));
    }
    computations[size - 1] = elems[size - 1];
    for (size_t i = size - 1; i-- > 0;) computations[i] = op == 32 ? ef_mul(elems[i], computations[i + 1]) : ef_add(elems[i], computations[i + 1]);
    const EF result = computations[0];
    memory.set_ef_element(to_usize(ptr_res), result);
    auto& c = tr.extension;
    for (size_t i = 0; i < size; i++) {
        const size_t current_len = size - i;
        c[0].push_back(in.is_be ? ONE : 0);
        c[1].push_back(i == 0 ? ONE : 0);
        c[3].push_back(op == 8 ? ONE : 0);
        c[4].push_back(op == 16 ? ONE : 0);
        c[5].push_back(op == 32 ? ONE : 0);
        c[2].push_back(from_usize(current_len));
        c[6].push_back(idx_as[i]);
        c[7].push_back(idx_bs[i]);
        c[13].push_back(ptr_res);
        for (int k = 0; k < 5; k++) {
            c[14 + k].push_back(0);  // VALUE_A: fill_trace_extension_op
            c[19 + k].push_back(v_bs[i].v[k]);
            c[24 + k].push_back(result.v[k]);
            c[8 + k].push_back(computations[i].v[k]);
        }
        c[29].push_back(i == 0 ? ONE : 0);
        c[30].push_back(from_usize(op + 4 * in.is_be + 64 * current_len));
    }
}

// ---- the machine ---------------------------------------------------------------------------------------------------------------------------
struct Cursor {
    std::vector<size_t> index;
};
struct ParallelBatchInfo {
    size_t batch_pc, batch_fp, frame_size, n_args;
    Operand end_value;
    std::vector<size_t> hint_indices_at_start;
};
enum class LoopExit { Halted, LoopBack, ParallelBatch };

static inline Operand hint_operand(const Hint& h, int k) {
    Operand o;
    o.kind = (Arg)h.mode[k];
    if (o.kind == Arg::Constant)
        o.constant = from_usize(h.args[k]);
    else
        o.offset = h.args[k];
    return o;
}

static inline void execute_hint(const Hint& h, MemoryAccess& memory, size_t fp, size_t& ap, const WitnessHints& w, Cursor& cur, Trace& tr) {
    auto val = [&](int k) { return read_value(hint_operand(h, k), memory, fp); };
    auto decompose_chunks = [&](size_t value, size_t chunk_size, size_t& at) {
        if (chunk_size == 0 || 24 % chunk_size) throw RunnerError("Panic: 24 is not a multiple of chunk_size");
        for (size_t i = 0; i < 24 / chunk_size; i++) memory.set(at++, from_usize((value >> (chunk_size * i)) & (((size_t)1 << chunk_size) - 1)));
    };
    switch (h.kind) {
        case 1: {  // Inverse { arg, res_offset }
            const F v = val(0);
            memory.set(fp + h.args[1], v ? inv(v) : 0);
            break;
        }
        case 2: {  // RequestMemory { offset, size }
            const size_t size = to_usize(val(1));
            memory.set(fp + h.args[0], from_usize(ap));
            ap += size;
            break;
        }
        case 3:  // DerefHint { offset_src, offset_target }
            tr.pending_deref_hints.emplace_back(fp + h.args[1], fp + h.args[0]);
            break;
        case 4: {  // DecomposeBitsXMSS
            size_t at = to_usize(val(0));
            const size_t src = to_usize(val(1)), num = to_usize(val(2)), chunk = to_usize(val(3));
            for (size_t i = 0; i < num; i++) decompose_chunks(to_usize(memory.get(src + i)), chunk, at);
            break;
        }
        case 5: {  // DecomposeBitsMerkleWhir
            size_t at = to_usize(val(0));
            decompose_chunks(to_usize(val(1)), to_usize(val(2)), at);
            break;
        }
        case 6: {  // DecomposeBits: big endian
            const size_t x = to_usize(val(0)), at = to_usize(val(1)), bits = to_usize(val(2));
            if (bits > 31) throw RunnerError("Panic: num_bits <= F::bits()");
            for (size_t j = 0; j < bits; j++) memory.set(at + j, ((x >> (bits - 1 - j)) & 1) ? ONE : 0);
            break;
        }
        case 7: {  // LessThan
            const F a = val(0), b = val(1);
            memory.set(memory_address(hint_operand(h, 2), fp), to_usize(a) < to_usize(b) ? ONE : 0);
            break;
        }
        case 8: {  // Log2Ceil
            const size_t n = to_usize(val(0));
            size_t l = 0;
            while (((size_t)1 << l) < n) l++;
            memory.set(memory_address(hint_operand(h, 1), fp), from_usize(l));
            break;
        }
        case 9:
        case 10: {  // HintWitness { name, Inline { offset } | Indirect { ptr_offset } }
            const size_t name = h.args[0];
            const size_t e = w.name_entry_begin[name] + cur.index[name];
            if (e >= w.name_entry_begin[name + 1]) throw RunnerError("Panic: hint_witness: exhausted entries");
            cur.index[name]++;
            size_t dest = h.kind == 9 ? fp + h.args[1] : to_usize(memory.get(fp + h.args[1]));
            for (uint64_t k = w.entry_offset[e]; k < w.entry_offset[e + 1]; k++) memory.set(dest++, w.data[k]);
            break;
        }
        case 12: {  // DebugAssert
            const size_t l = to_usize(val(0)), r = to_usize(val(1));
            if (h.args[3] && r >= ((size_t)1 << MIN_LOG_MEMORY_SIZE)) throw RunnerError("RangeCheckWithTooBigRange");
            const bool ok = h.args[2] == 0 ? l == r : h.args[2] == 1 ? l != r : h.args[2] == 2 ? l < r : l <= r;
            if (!ok) throw RunnerError("DebugAssertFailed");
            break;
        }
        default: break;  // 11 ParallelBatchStart: handled by run_loop
    }
}

static inline void execute_instruction(const Instruction& in, MemoryAccess& memory, size_t& pc, size_t& fp, Trace& tr) {
    switch (in.op) {
        case Op::Add:
        case Op::Mul: {
            const bool is_mul = in.op == Op::Mul;
            auto compute = [&](F a, F b) { return is_mul ? mul(a, b) : add(a, b); };
            auto inverse_compute = [&](F a, F b) {
                if (!is_mul) return sub(a, b);
                if (b == 0) throw RunnerError("DivByZero");
                return mul(a, inv(b));
            };
            const Operand &arg_a = in.a, &res = in.b, &arg_c = in.c;
            if (is_value_unknown(res, memory, fp)) {
                const size_t at = memory_address(res, fp);
                const F a = read_value(arg_a, memory, fp), b = read_value(arg_c, memory, fp);
                memory.set(at, compute(a, b));
            } else if (is_value_unknown(arg_a, memory, fp)) {
                const size_t at = memory_address(arg_a, fp);
                const F r = read_value(res, memory, fp), b = read_value(arg_c, memory, fp);
                memory.set(at, inverse_compute(r, b));
            } else if (is_value_unknown(arg_c, memory, fp)) {
                const size_t at = memory_address(arg_c, fp);
                const F r = read_value(res, memory, fp), a = read_value(arg_a, memory, fp);
                memory.set(at, inverse_compute(r, a));
            } else {
                const F a = read_value(arg_a, memory, fp), b = read_value(arg_c, memory, fp), r = read_value(res, memory, fp);
                if (r != compute(a, b)) throw RunnerError("NotEqual");
            }
            (is_mul ? tr.mul : tr.add)++;
            pc++;
            break;
        }
        case Op::Deref: {
            const size_t shift_0 = in.a.offset, shift_1 = to_usize(in.b.constant);
            if (is_value_unknown(in.c, memory, fp)) {
                const size_t at = memory_address(in.c, fp);
                const F ptr = memory.get(fp + shift_0);
                if (auto v = memory.try_get(to_usize(ptr) + shift_1)) memory.set(at, *v);
            } else {
                const F value = read_value(in.c, memory, fp);
                const F ptr = memory.get(fp + shift_0);
                memory.set(to_usize(ptr) + shift_1, value);
            }
            tr.deref++;
            pc++;
            break;
        }
        case Op::Jump: {
            const F cond = read_value(in.a, memory, fp);
            if (cond != 0 && cond != ONE) throw RunnerError("Panic: jump condition is not boolean");
            if (cond == 0)
                pc++;
            else {
                const size_t new_pc = to_usize(read_value(in.b, memory, fp));
                fp = to_usize(read_value(in.c, memory, fp));
                pc = new_pc;
            }
            tr.jump++;
            break;
        }
        default: {
            const F a = read_value(in.a, memory, fp), b = read_value(in.b, memory, fp), c = read_value(in.c, memory, fp);
            if (in.op == Op::Poseidon16)
                poseidon16_execute(in, a, b, c, memory, tr);
            else
                ext_exec_multi_row(in, a, b, c, memory, tr);
            pc++;
        }
    }
}

static inline LoopExit run_loop(const Bytecode& bc, MemoryAccess& memory, Trace& trace, size_t& pc, size_t& fp, size_t& ap, const WitnessHints& w,
                                Cursor& cur, std::optional<size_t> stop_pc, std::optional<ParallelBatchInfo>& out_batch) {
    std::optional<ParallelBatchInfo> parallel_batch;
    for (;;) {
        if (pc == bc.ending_pc) return LoopExit::Halted;
        if (pc >= bc.code.size()) throw RunnerError("PCOutOfBounds");
        trace.pcs.push_back(pc);
        trace.fps.push_back(fp);
        for (const Hint& h : bc.hints[pc]) {
            if (h.kind == 11) {
                if (!parallel_batch) parallel_batch = ParallelBatchInfo{pc, fp, ap - fp, h.args[0], hint_operand(h, 1), cur.index};
                continue;
            }
            execute_hint(h, memory, fp, ap, w, cur, trace);
        }
        execute_instruction(bc.code[pc], memory, pc, fp, trace);
        if (stop_pc && *stop_pc == pc) return LoopExit::LoopBack;
        if (parallel_batch && pc == parallel_batch->batch_pc) {
            out_batch = parallel_batch;
            return LoopExit::ParallelBatch;
        }
    }
}

static inline void resolve_deref_hints(Memory& memory, const std::vector<std::pair<size_t, size_t>>& pending) {
    std::set<size_t> resolved;
    for (;;) {
        bool made_progress = false;
        for (auto [target_addr, src_addr] : pending) {
            if (resolved.count(target_addr)) continue;
            const F addr = memory.get(src_addr);
            auto value = memory.try_get(to_usize(addr));
            if (!value) continue;
            memory.set(target_addr, *value);
            resolved.insert(target_addr);
            made_progress = true;
        }
        if (!made_progress) break;
    }
    for (auto [target_addr, src_addr] : pending) {
        (void)src_addr;
        if (!resolved.count(target_addr)) memory.set(target_addr, 0);
    }
}

static inline void handle_parallel_batch(const Bytecode& bc, Memory& memory, Trace& trace, const WitnessHints& w, Cursor& cur, size_t& pc, size_t& fp,
                                         size_t& ap, const ParallelBatchInfo& batch) {
    const size_t start_value = to_usize(memory.get(batch.batch_fp + 2));
    const size_t end_value = to_usize(read_value(batch.end_value, memory, batch.batch_fp));
    if (end_value < start_value || end_value == start_value) throw RunnerError("Panic: parallel batch bounds");
    const size_t n_iters = end_value - start_value;
    if (n_iters == 1) return;
    const size_t stride = fp - batch.batch_fp;
    const F return_pc = memory.get(fp), saved_fp = memory.get(fp + 1);
    std::vector<F> args;
    for (size_t i = 0; i < batch.n_args; i++) args.push_back(memory.get(batch.batch_fp + 2 + i));
    std::vector<size_t> named_per_iter(cur.index.size());
    for (size_t k = 0; k < cur.index.size(); k++) named_per_iter[k] = cur.index[k] - batch.hint_indices_at_start[k];
    for (size_t i = 1; i <= n_iters; i++) {  // write_call_frame
        const size_t f = batch.batch_fp + i * stride, iter_val = i < n_iters ? start_value + i : end_value;
        memory.set(f, from_usize(to_usize(return_pc)));
        memory.set(f + 1, from_usize(to_usize(saved_fp)));
        memory.set(f + 2, from_usize(iter_val));
        for (size_t j = 1; j < args.size(); j++) memory.set(f + 2 + j, args[j]);
    }
    const size_t max_addr = batch.batch_fp + (n_iters + 1) * stride;
    if (max_addr > memory.cells.size()) memory.cells.resize(max_addr);
    const size_t n_par = n_iters - 1, split_at = batch.batch_fp + stride;
    std::vector<Trace> seg_traces(n_par);
    std::vector<std::vector<std::pair<size_t, F>>> seg_deferred(n_par);
    for (size_t i = 0; i < n_par; i++) {
        SegmentMemory seg;
        seg.all = &memory.cells;
        seg.shared_len = split_at;
        seg.segment_start = split_at + i * stride;
        seg.segment_len = stride;
        seg.segment.assign(memory.cells.begin() + seg.segment_start, memory.cells.begin() + seg.segment_start + stride);
        size_t seg_pc = batch.batch_pc, seg_fp = batch.batch_fp + (i + 1) * stride, seg_ap = seg_fp + batch.frame_size;
        Cursor seg_cur = cur;
        for (size_t k = 0; k < seg_cur.index.size(); k++) seg_cur.index[k] += i * named_per_iter[k];
        std::optional<ParallelBatchInfo> inner;
        try {
            if (run_loop(bc, seg, seg_traces[i], seg_pc, seg_fp, seg_ap, w, seg_cur, batch.batch_pc, inner) != LoopExit::LoopBack)
                throw RunnerError("Panic: segment did not loop back");
        } catch (const RunnerError& e) {
            throw RunnerError("ParallelSegmentFailed(" + std::to_string(i + 1) + ", " + e.what() + ")");
        }
        // the reference's segments write into disjoint slices of the one memory: copy this one back
        std::copy(seg.segment.begin(), seg.segment.end(), memory.cells.begin() + seg.segment_start);
        seg_deferred[i] = std::move(seg.deferred_writes);
    }
    for (size_t i = 0; i < n_par; i++) {
        trace.merge(std::move(seg_traces[i]));
        for (auto [addr, val] : seg_deferred[i]) memory.set(addr, val);
    }
    for (size_t k = 0; k < cur.index.size(); k++) cur.index[k] += n_par * named_per_iter[k];
    pc = batch.batch_pc;
    fp = batch.batch_fp + n_iters * stride;
    ap = fp + batch.frame_size;
}

struct ExecutionResult {
    Memory memory;
    Trace trace;
    size_t public_memory_size = 0, runtime_memory_size = 0;
};

static inline ExecutionResult execute_bytecode(const Bytecode& bc, const uint32_t* public_input, size_t n_public_input, const WitnessHints& w) {
    ExecutionResult r;
    size_t pub = 1;  // padd_with_zero_to_next_power_of_two: 0usize.next_power_of_two() == 1, an empty public input is one zero word
    while (pub < n_public_input) pub <<= 1;
    r.memory.cells.assign(pub, F(0));
    for (size_t i = 0; i < n_public_input; i++) r.memory.cells[i] = public_input[i];
    size_t fp = pub + w.preamble_memory_len;
    fp = (fp + 4) / 5 * 5;
    const size_t initial_ap = fp + bc.starting_frame_memory;
    size_t pc = 0, ap = initial_ap;
    Cursor cur;
    cur.index.assign(bc.n_names, 0);
    for (;;) {
        std::optional<ParallelBatchInfo> batch;
        const LoopExit e = run_loop(bc, r.memory, r.trace, pc, fp, ap, w, cur, std::nullopt, batch);
        if (e == LoopExit::Halted) break;
        handle_parallel_batch(bc, r.memory, r.trace, w, cur, pc, fp, ap, *batch);
    }
    resolve_deref_hints(r.memory, r.trace.pending_deref_hints);
    for (size_t k = 0; k < bc.n_names; k++)
        if (cur.index[k] != w.name_entry_begin[k + 1] - w.name_entry_begin[k]) throw RunnerError("Panic: not all entries of a named hint were consumed");
    r.trace.pcs.push_back(pc);
    r.trace.fps.push_back(fp);
    r.public_memory_size = pub;
    r.runtime_memory_size = ap - initial_ap;
    return r;
}

// ---- get_execution_trace (lean_prover/src/trace_gen.rs:14-191) ----------------------------------------------------------------------------
struct ExecutionTrace {
    std::vector<F> memory;                       // padded
    std::vector<std::vector<F>> tables[3];       // execution 24, extension_op 31, poseidon16 111 columns, 2^log_n_rows each
    size_t non_padded_n_rows[3], log_n_rows[3];
    size_t zero_vec_ptr, null_hash_ptr;
};

static inline ExecutionTrace get_execution_trace(const Bytecode& bc, ExecutionResult& er) {
    ExecutionTrace t;
    const size_t n_cycles = er.trace.pcs.size();
    for (auto& c : er.memory.cells) t.memory.push_back(c.value_or(0));
    t.zero_vec_ptr = t.memory.size();
    t.memory.insert(t.memory.end(), 16, 0);
    t.null_hash_ptr = t.memory.size();
    {
        F z[16] = {0};
        poseidon16_compress(z);
        t.memory.insert(t.memory.end(), z, z + 8);
    }
    size_t padded = std::max<size_t>(std::max(t.memory.size(), n_cycles), (size_t)1 << MIN_LOG_N_ROWS_PER_TABLE);
    size_t p2 = 1;
    while (p2 < padded) p2 <<= 1;
    t.memory.resize(p2, 0);
    auto mem = [&](F addr) -> F {
        const size_t a = to_usize(addr);
        return a < er.memory.cells.size() ? er.memory.cells[a].value_or(0) : 0;  // memory.0.get(addr).flatten().unwrap_or_default()
    };
    auto& ex = t.tables[0];
    ex.assign(24, std::vector<F>(n_cycles));
    const F TWO = add(ONE, ONE);
    for (size_t i = 0; i < n_cycles; i++) {
        const size_t pc = er.trace.pcs[i];
        const F fp = from_usize(er.trace.fps[i]);
        const uint32_t* f = bc.instructions_multilinear + 16 * pc;
        const F operand_a = f[0], operand_b = f[1], operand_c = f[2], flag_a = f[3], flag_b = f[4], flag_c = f[5], flag_c_fp = f[6], flag_ab_fp = f[7];
        const bool is_deref = f[10] == TWO;
        F addr_a = 0;
        if (flag_a == 0 && flag_ab_fp == 0) addr_a = add(fp, operand_a);
        const F value_a = mem(addr_a);
        F addr_b = 0;
        if (flag_b == 0 && flag_ab_fp == 0)
            addr_b = add(fp, operand_b);
        else if (is_deref)
            addr_b = add(value_a, operand_b);
        const F value_b = mem(addr_b);
        F addr_c = 0;
        if (flag_c == 0 && flag_c_fp == 0) addr_c = add(fp, operand_c);
        const F value_c = mem(addr_c);
        for (int j = 0; j < 12; j++) ex[8 + j][i] = f[j];
        auto nu = [&](F flag, F flag_fp, F operand, F value) {
            return add(add(mul(flag, operand), mul(sub(sub(ONE, flag), flag_fp), value)), mul(flag_fp, add(fp, operand)));
        };
        const Op op = bc.code[pc].op;
        ex[20][i] = (op == Op::Poseidon16 || op == Op::ExtensionOp) ? ONE : 0;
        ex[21][i] = nu(flag_a, flag_ab_fp, operand_a, value_a);
        ex[22][i] = nu(flag_b, flag_ab_fp, operand_b, value_b);
        ex[23][i] = nu(flag_c, flag_c_fp, operand_c, value_c);
        ex[5][i] = value_a, ex[6][i] = value_b, ex[7][i] = value_c;
        ex[0][i] = from_usize(pc), ex[1][i] = fp;
        ex[2][i] = addr_a, ex[3][i] = addr_b, ex[4][i] = addr_c;
    }
    // Poseidon table: fill_trace_poseidon_16 + the output override for permute = 0 rows
    auto& pos = t.tables[2];
    pos = er.trace.poseidon;
    const size_t n_pos = pos[0].size();
    for (int c = 25; c < 109; c++) pos[c].assign(n_pos, 0);
    for (size_t i = 0; i < n_pos; i++) {
        F row[109];
        for (int c = 0; c < 109; c++) row[c] = pos[c][i];
        poseidon16_fill_row(row);
        if (row[8] == 0) {
            const size_t base = to_usize(row[2]);
            if (row[3] == ONE)
                for (int j = 0; j < 4; j++) row[97 + j] = t.memory[base + 4 + j];
            for (int j = 0; j < 8; j++) row[101 + j] = t.memory[base + 8 + j];
        }
        for (int c = 25; c < 109; c++) pos[c][i] = row[c];
    }
    // ExtensionOp: fill_trace_extension_op
    auto& ext = t.tables[1];
    ext = er.trace.extension;
    for (size_t i = 0; i < ext[6].size(); i++) {
        const size_t addr = to_usize(ext[6][i]);
        for (int k = 0; k < 5; k++) ext[14 + k][i] = t.memory[addr + k];
    }
    // pad_table
    for (int tb = 0; tb < 3; tb++) {
        auto& cols = t.tables[tb];
        const size_t h = cols[0].size();
        t.non_padded_n_rows[tb] = h;
        size_t l = 0;
        while (((size_t)1 << l) < h + 1) l++;
        t.log_n_rows[tb] = std::max(l, MIN_LOG_N_ROWS_PER_TABLE);
        const size_t n_rows = (size_t)1 << t.log_n_rows[tb];
        std::vector<F> row(cols.size(), 0);
        if (tb == 0) {  // execution/mod.rs:59-74
            row[0] = from_usize(bc.ending_pc);
            row[17] = ONE, row[11] = ONE, row[8] = ONE, row[12] = ONE;
            row[9] = from_usize(bc.ending_pc);
            row[14] = ONE;
            row[21] = ONE;
            row[22] = from_usize(bc.ending_pc);
            row[2] = row[3] = row[4] = from_usize(t.zero_vec_ptr);
        } else if (tb == 1) {  // extension_op/mod.rs:125-134
            row[1] = ONE, row[2] = ONE;
            row[30] = from_usize(64);
            row[6] = row[7] = row[13] = from_usize(t.zero_vec_ptr);
        } else {  // poseidon_16/mod.rs:182-205
            F r109[109] = {0};
            r109[1] = from_usize(t.zero_vec_ptr);
            r109[2] = from_usize(t.null_hash_ptr);
            r109[6] = from_usize(t.zero_vec_ptr);
            r109[7] = from_usize(t.zero_vec_ptr + 4);
            poseidon16_fill_row(r109);
            for (int c = 0; c < 109; c++) row[c] = r109[c];
            row[109] = from_usize(t.zero_vec_ptr);
            row[110] = from_usize(1);
        }
        for (size_t c = 0; c < cols.size(); c++) cols[c].resize(n_rows, row[c]);
    }
    return t;
}

}  // namespace vm
}  // namespace orc
