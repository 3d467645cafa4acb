// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).
//
// Numerator / denominator construction of prove_generic_logup (crates/sub_protocols/src/logup.rs:27-201) in natural
// index order (the reference additionally bit-reverses 2^12-chunks and SIMD-packs, logup.rs:61-86 — a CPU layout).
#pragma once
#include "air_oracle.hpp"

namespace orc {

struct LookupIntoMemory {
    size_t index;
    std::vector<size_t> values;
};
struct BusDef {
    bool pull;
    size_t selector;
    std::vector<size_t> data;
};
struct VmTableDef {
    size_t n_columns_total;
    std::vector<LookupIntoMemory> lookups;
    BusDef bus;
};
static inline std::vector<size_t> range(size_t a, size_t b) {
    std::vector<size_t> r;
    for (size_t i = a; i < b; i++) r.push_back(i);
    return r;
}
// lean_vm/src/tables/execution/mod.rs:29-60, extension_op/mod.rs:90-123, poseidon_16/mod.rs:126-174
static inline VmTableDef vm_table_def(int t) {
    if (t == AIR_EXECUTION) return {24, {{2, {5}}, {3, {6}}, {4, {7}}}, {false, 20, {19, 21, 22, 23}}};
    if (t == AIR_EXTENSION_OP) return {31, {{6, range(14, 19)}, {7, range(19, 24)}, {13, range(24, 29)}}, {true, 29, {30, 6, 7, 13}}};
    return {111, {{6, range(9, 13)}, {7, range(13, 17)}, {1, range(17, 25)}, {2, range(93, 109)}}, {true, 0, {110, 109, 1, 2}}};
}

struct VmTableTrace {
    int table;
    size_t log_rows;
    const uint32_t* cols;  // n_columns_total x 2^log_rows, column major
    const uint32_t* col(size_t c) const { return cols + (c << log_rows); }
};

// finger_print (utils/src/multilinear.rs:76-85): sum_j alpha[j] data[j] + alpha[15] * domsep
static inline EF finger_print(uint32_t domsep, const std::vector<uint32_t>& data, const EF* alphas) {
    EF s = ef_mul_base(alphas[15], to_monty(domsep));
    for (size_t j = 0; j < data.size(); j++) s = ef_add(s, ef_mul_base(alphas[j], data[j]));
    return s;
}

// tables must be sorted by descending height (sort_tables_by_height).  Returns total_active_len; nums/dens are padded
// with (0, 1) to the next power of two.
static inline size_t logup_fill(const uint32_t* memory, const uint32_t* memory_acc, size_t log_mem, const uint32_t* bytecode,
                                const uint32_t* bytecode_acc, size_t log_bytecode, const std::vector<VmTableTrace>& tables, EF c,
                                const EF* alphas, std::vector<uint32_t>& nums, std::vector<EF>& dens) {
    nums.clear();
    dens.clear();
    auto push = [&](uint32_t n, EF d) {
        nums.push_back(n);
        dens.push_back(d);
    };
    const size_t max_h = (size_t)1 << tables[0].log_rows;
    for (size_t i = 0; i < ((size_t)1 << log_mem); i++)  // logup.rs:94-109
        push(neg(memory_acc[i]), ef_sub(c, finger_print(0, {memory[i], to_monty((uint32_t)i)}, alphas)));
    for (size_t i = 0; i < ((size_t)1 << log_bytecode); i++) {  // :111-125
        std::vector<uint32_t> d;
        for (int k = 0; k < 12; k++) d.push_back(bytecode[i * 16 + k]);
        d.push_back(to_monty((uint32_t)i));
        push(neg(bytecode_acc[i]), ef_sub(c, finger_print(2, d, alphas)));
    }
    for (size_t i = (size_t)1 << log_bytecode; i < max_h; i++) push(0, ef_one());  // :126-135
    for (const VmTableTrace& t : tables) {
        const size_t n = (size_t)1 << t.log_rows;
        const VmTableDef def = vm_table_def(t.table);
        if (t.table == AIR_EXECUTION) {  // :141-156
            for (size_t r = 0; r < n; r++) {
                std::vector<uint32_t> d;
                for (int k = 0; k < 12; k++) d.push_back(t.col(8 + k)[r]);
                d.push_back(t.col(0)[r]);
                push(ONE, ef_sub(c, finger_print(2, d, alphas)));
            }
        }
        for (size_t r = 0; r < n; r++) {  // bus, :158-176
            std::vector<uint32_t> d;
            for (size_t col : def.bus.data) d.push_back(t.col(col)[r]);
            uint32_t sel = t.col(def.bus.selector)[r];
            push(def.bus.pull ? neg(sel) : sel, ef_add(c, finger_print(1, d, alphas)));
        }
        for (const LookupIntoMemory& lk : def.lookups)  // :178-199
            for (size_t i = 0; i < lk.values.size(); i++)
                for (size_t r = 0; r < n; r++)
                    push(ONE, ef_sub(c, finger_print(0, {t.col(lk.values[i])[r], add(t.col(lk.index)[r], to_monty((uint32_t)i))}, alphas)));
    }
    size_t total = nums.size(), p2 = 1;
    while (p2 < total) p2 <<= 1;
    while (nums.size() < p2) push(0, ef_one());
    return total;
}

}  // namespace orc
