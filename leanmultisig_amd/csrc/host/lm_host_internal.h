// Internal to the host layer (lm_host.cpp, lm_wire.cpp, lm_verify.cpp): the transcript objects behind the opaque handles of
// include/leanmultisig_host.h.  Not part of the ABI.
#pragma once
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../../include/leanmultisig_host.h"
#include "../kb.h"
#include "../poseidon16.h"

void lm_set_error(const char* fmt, ...);  // lm_core.hip (thread-local message behind lm_last_error)
// lm_core.hip: per-context cache of device copies of long-lived host objects, keyed by the object's process-unique id (+ a small
// tag); the entries are pool allocations of the context (lm_malloc) and die with it
unsigned long long lm_ctx_uid(lm_ctx* ctx);         // process-unique id of a live context
lm_ctx* lm_ctx_by_uid(unsigned long long uid);      // nullptr once that context has been destroyed
// fn(ctx, arg) under the registry lock iff context `uid` is alive and is `expect` (the context cannot be destroyed meanwhile)
bool lm_ctx_with_live(unsigned long long uid, lm_ctx* expect, void (*fn)(lm_ctx*, void*), void* arg);
void* lm_ctx_cache_get(lm_ctx* ctx, unsigned long long key);
void lm_ctx_cache_put(lm_ctx* ctx, unsigned long long key, void* p);

namespace lmh {
using kb::EF;
using kb::u32;
using kb::u64;

// ---- Challenger (crates/backend/fiat-shamir/src/challenger.rs:9-76): overwrite-mode duplex, plain permutation -----
// Poseidon1-16 permutation on the host: AVX-512 when the CPU has it, else the scalar code (lm_poseidon_x86.cpp)
void host_permute(u32 state[16]);
inline void host_compress(u32 s[16]) {  // perm(x) + x (poseidon1_koalabear_16.rs:1018-1030)
    u32 in[16];
    memcpy(in, s, sizeof in);
    host_permute(s);
    for (int i = 0; i < 16; i++) s[i] = kb::add(s[i], in[i]);
}

struct Challenger {
    u32 state[16];
    bool rate_fresh = false;
    Challenger() { memset(state, 0, sizeof state); }
    void observe(const u32 v[8]) {
        memcpy(state + 8, v, 32);
        host_permute(state);
        rate_fresh = true;
    }
    void observe_many(const u32* s, u64 n) {
        for (u64 off = 0; off < n; off += 8) {
            u32 buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            memcpy(buf, s + off, (size_t)std::min<u64>(8, n - off) * 4);
            observe(buf);
        }
    }
    void duplex() {
        const u32 z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        observe(z);
    }
    bool sample(u32 out[8]) {
        if (!rate_fresh) return false;  // "stale rate. insert a duplex() before."
        memcpy(out, state + 8, 32);
        rate_fresh = false;
        return true;
    }
    bool sample_many(u64 n_blocks, std::vector<u32>& out) {
        out.clear();
        for (u64 i = 0; i < n_blocks; i++) {
            if (i) duplex();
            u32 b[8];
            if (!sample(b)) return false;
            out.insert(out.end(), b, b + 8);
        }
        return true;
    }
};


struct Opening {  // one Merkle opening as the prover produced it (MerkleOpening, fiat-shamir/src/transcript.rs:8-12)
    u64 index;
    std::vector<u32> leaf, path;
};

// PrunedMerklePaths (fiat-shamir/src/merkle_pruning.rs:5-12): the openings of one query set after MerklePaths::prune
struct PrunedPath {
    u64 leaf_index;
    std::vector<u32> leaf;      // without the common zero tail
    std::vector<u32> siblings;  // 8 words per kept sibling, bottom-up
};
struct PrunedBatch {
    u32 merkle_height = 0, n_trailing_zeros = 0;
    std::vector<u32> original_order;
    std::vector<PrunedPath> paths;
};
}  // namespace lmh

struct lmh_prover {
    lmh::Challenger ch;
    std::vector<lmh::u32> transcript;
    std::vector<lmh::Opening> openings;
    std::vector<lmh::u32> batch_sizes;  // openings per hint_merkle_paths call (one query set of one commitment), in order
    // the pruned blob of the current state (size query + copy are two calls): valid while the three sizes are unchanged
    double stage_ms[LMH_N_STAGES] = {0, 0, 0, 0, 0, 0, 0, 0};  // wall clock of the last lmh_prove_execution per stage (lmh_prover_stage_times)
    mutable std::vector<lmh::u32> pruned_cache;
    mutable size_t pruned_key[3] = {~(size_t)0, 0, 0};
};

namespace lmh {
std::vector<PrunedBatch> prune(const lmh_prover* p);  // MerklePaths::prune per batch (lm_host.cpp)
// lmh_pad_table; with_virtual: d_cols has n_total entries and the virtual bus columns of the padding rows are filled as well
int pad_table(lm_ctx* ctx, uint32_t table, uint32_t* const* d_cols, uint64_t n_rows, uint32_t log_rows, uint32_t zero_vec_ptr, uint32_t null_hash_ptr,
              uint32_t ending_pc, bool with_virtual);

// ---- leanVM table metadata the prover needs (bus and memory lookups) ------------------------------------------------
// lean_vm/src/tables/execution/mod.rs:29-60, extension_op/mod.rs:90-123, poseidon_16/mod.rs:126-174
struct VmLookup {
    u32 index, first_value, n_values;
};
struct VmTableDef {
    u32 n_columns, n_shift, n_total;
    u32 n_lookups;
    VmLookup lookups[4];
    bool pull;
    u32 selector, bus_data[4];
};
static const VmTableDef kVmTables[3] = {
    {20, 2, 24, 3, {{2, 5, 1}, {3, 6, 1}, {4, 7, 1}, {0, 0, 0}}, false, 20, {19, 21, 22, 23}},
    {29, 13, 31, 3, {{6, 14, 5}, {7, 19, 5}, {13, 24, 5}, {0, 0, 0}}, true, 29, {30, 6, 7, 13}},
    {109, 0, 111, 4, {{6, 9, 4}, {7, 13, 4}, {1, 17, 8}, {2, 93, 16}}, true, 0, {110, 109, 1, 2}},
};
static const u32 kSnarkDomainSep[8] = {130704175, 1303721200, 493664240, 1035493700,
                                2063844858, 1410214009, 1938905908, 1696767928};  // lean_prover/src/lib.rs:30-32
inline u32 log2_ceil_u64(u64 x) {
    u32 l = 0;
    while ((1ull << l) < x) l++;
    return l;
}
inline void sorted_tables(const u32 log_rows[3], int order[3]) {  // sort_tables_by_height (stable, descending)
    order[0] = 0;
    order[1] = 1;
    order[2] = 2;
    std::stable_sort(order, order + 3, [&](int a, int b) { return log_rows[a] > log_rows[b]; });
}

// lean_vm/src/core/constants.rs:13-37
static constexpr u32 MIN_LOG_MEMORY_SIZE = 16, MAX_LOG_MEMORY_SIZE = 26, MIN_BYTECODE_LOG_SIZE = 8, MIN_LOG_N_ROWS_PER_TABLE = 8;
inline u32 max_log_n_rows_per_table(int table) { return table == 0 ? 24 : 21; }
inline bool rate_ok(u32 log_inv_rate) { return log_inv_rate >= 1 && log_inv_rate <= 4; }  // check_rate, lean_prover/src/lib.rs:52-58

// The flattened statement list of lmh_whir_prove / the verifier (SparseStatement, crates/whir/src/lib.rs:31-108)
struct Statements {
    std::vector<lm_sparse_statement> sts;
    std::vector<u32> pts, vals;  // EF coordinates / values, 5 words each
    std::vector<u64> sels;
    void begin(const u32* point, u32 point_len, u32 is_next) {
        lm_sparse_statement s;
        memset(&s, 0, sizeof s);
        s.point_len = point_len;
        s.is_next = is_next;
        s.point_offset = pts.size() / 5;
        s.values_offset = sels.size();
        if (point_len) pts.insert(pts.end(), point, point + (size_t)point_len * 5);
        sts.push_back(s);
    }
    void value(u64 selector, const EF& v) {
        sels.push_back(selector);
        vals.insert(vals.end(), v.v, v.v + 5);
        sts.back().n_values++;
    }
};
struct ColVal {
    u32 col;
    EF v;
};
// previous_statements of prove_execution.rs:233-256 followed by stacked_pcs_global_statements (sub_protocols/src/stacked_pcs.rs:40-97):
// gkr_point: the GKR claim point (gkr_n_vars x 5 words); columns_values[t]: the logup column evaluations of table t;
// air_point: n_max challenges of the batched AIR sumcheck (5 words each, sumcheck order); col_evals: per table in sorted
// order, (n_columns + n_shift) x 5 words.
inline void assemble_statements(Statements& S, const u32 log_rows[3], u32 log_mem, u32 log_bc, u32 ending_pc, const u32* gkr_point,
                                u32 gkr_n_vars, const EF& value_memory, const EF& value_memory_acc, const EF& value_bytecode_acc,
                                const u32* pm_point, u32 lpm, const EF& pm_eval, const std::vector<ColVal> columns_values[3],
                                const u32* air_point, const u32* col_evals) {
    int order[3];
    sorted_tables(log_rows, order);
    auto from_end = [&](u32 n) { return gkr_point + (size_t)(gkr_n_vars - n) * 5; };
    const u64 mem = 1ull << log_mem;
    S.begin(from_end(log_mem), log_mem, 0);
    S.value(0, value_memory);
    S.value(1, value_memory_acc);
    S.begin(pm_point, lpm, 0);
    S.value(0, pm_eval);
    S.begin(from_end(log_bc), log_bc, 0);
    S.value((2 * mem) >> log_bc, value_bytecode_acc);
    u64 soff = 2 * mem + (1ull << std::max(log_bc, log_rows[order[0]]));
    const u32 n_max = log_rows[order[0]];
    const u32* ce = col_evals;
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const VmTableDef& def = kVmTables[t];
        const u32 nv = log_rows[t];
        if (t == 0) {
            S.begin(nullptr, 0, 0);  // unique_value(STARTING_PC = 0)
            S.value(soff + (0ull << nv), kb::ef_zero());
            S.begin(nullptr, 0, 0);
            S.value(soff + (1ull << nv) - 1, kb::ef_from_base(kb::to_monty(ending_pc)));
        }
        // first committed statement: logup column values at from_end(gkr_point, nv), ascending column index (BTreeMap)
        std::vector<ColVal> cvs = columns_values[t];
        std::sort(cvs.begin(), cvs.end(), [](const ColVal& a, const ColVal& b) { return a.col < b.col; });
        S.begin(from_end(nv), nv, 0);
        for (const ColVal& c : cvs) S.value((soff >> nv) + c.col, c.v);
        // second: AIR point (natural_ordering_point_for_session: last nv challenges reversed), next values then eq values
        std::vector<u32> nat((size_t)nv * 5);
        for (u32 j = 0; j < nv; j++) memcpy(&nat[5 * j], &air_point[(size_t)(n_max - 1 - j) * 5], 20);
        if (def.n_shift) {
            S.begin(nat.data(), nv, 1);
            for (u32 c = 0; c < def.n_shift; c++) {
                EF v;
                memcpy(v.v, ce + (size_t)(def.n_columns + c) * 5, 20);
                S.value((soff >> nv) + c, v);
            }
        }
        S.begin(nat.data(), nv, 0);
        for (u32 c = 0; c < def.n_columns; c++) {
            EF v;
            memcpy(v.v, ce + (size_t)c * 5, 20);
            S.value((soff >> nv) + c, v);
        }
        ce += (size_t)(def.n_columns + def.n_shift) * 5;
        soff += (u64)def.n_columns << nv;
    }
}
}  // namespace lmh
