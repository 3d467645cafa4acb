// aggregate_type_1 (crates/rec_aggregation/src/type_1_aggregation.rs:206-377) for raw signatures: everything the reference does inside
// its benchmark's stopwatch (rec_aggregation/src/benchmark.rs:397-410) BEFORE prove_execution — sort + dedup of the (public key,
// signature) pairs, hash_pubkeys, compute_tweak_table + its hash, build_type1_input_data + the public-input hash, the named hint
// streams — and the whole function on top of lmh_prove_execution_vm.  Host-only (the build compiles every source as HIP).
#if !defined(__HIP_DEVICE_COMPILE__)
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "lm_host_internal.h"
#include "lm_vm_internal.h"

using namespace lmh;

namespace {
// ---- constants of the program's memory layout (rec_aggregation/src/compilation.rs:18-26,43-68, xmss/src/lib.rs:19-41) ---------------
constexpr u32 DIGEST_LEN = 8, DIMENSION = 5, V = LM_XMSS_V, CHAIN_LENGTH = 8, LOG_LIFETIME = LM_XMSS_LOG_LIFETIME, XMSS_DIGEST_LEN = 4;
constexpr u32 RANDOMNESS_LEN = 6, PUB_KEY_FLAT_SIZE = 8, WOTS_SIG_SIZE_FE = RANDOMNESS_LEN + V * XMSS_DIGEST_LEN, MESSAGE_LEN_FE = 8;
constexpr u32 TWEAK_TYPE_CHAIN = 0, TWEAK_TYPE_WOTS_PK = 1, TWEAK_TYPE_MERKLE = 2, TWEAK_TYPE_ENCODING = 3;
constexpr u32 N_TWEAKS = 1 + V * CHAIN_LENGTH + 1 + LOG_LIFETIME, TWEAK_SLOT_SIZE = 4;
constexpr u32 TWEAK_TABLE_SIZE_FE_PADDED = (N_TWEAKS * TWEAK_SLOT_SIZE + DIGEST_LEN - 1) / DIGEST_LEN * DIGEST_LEN;
constexpr u32 ZERO_VEC_LEN = 16, NUM_REPEATED_ONES = 32;
constexpr u32 PREAMBLE_MEMORY_LEN = ZERO_VEC_LEN + DIGEST_LEN + DIMENSION + NUM_REPEATED_ONES + TWEAK_TABLE_SIZE_FE_PADDED;
constexpr u32 N_MERKLE_CHUNKS_FOR_SLOT = LOG_LIFETIME / 4, MAX_XMSS_AGGREGATED = 1u << 15, TYPE1_FLAG = 1;
constexpr u32 N_INSTRUCTION_COLUMNS_LOG = 4;  // log2_ceil(N_INSTRUCTION_COLUMNS = 12)
// lean_prover/src/lib.rs:30-32 (canonical)
constexpr u32 SNARK_DOMAIN_SEP[8] = {130704175, 1303721200, 493664240, 1035493700, 2063844858, 1410214009, 1938905908, 1696767928};
static_assert(LM_XMSS_SIG_WORDS == PUB_KEY_FLAT_SIZE + WOTS_SIG_SIZE_FE + LOG_LIFETIME * XMSS_DIGEST_LEN, "lm_xmss_sig layout");

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// poseidon_compress_slice (utils/src/poseidon.rs:41-67); n is a multiple of 8
void compress_slice(const u32* data, u64 n, bool use_iv, u32 out[8]) {
    alignas(64) u32 st[16];
    u64 at = 0;
    if (use_iv)
        memset(st, 0, 32);
    else if (n <= 16) {
        memset(st, 0, sizeof st);
        memcpy(st, data, n * 4);
        host_compress(st);
        memcpy(out, st, 32);
        return;
    } else {
        memcpy(st, data, 64);
        host_compress(st);
        at = 16;
    }
    for (; at < n; at += 8) {
        memcpy(st + 8, data + at, 32);
        host_compress(st);
    }
    memcpy(out, st, 32);
}
// make_tweak (xmss/src/lib.rs:43-53) as Montgomery words
inline void make_tweak(u32 type, u32 sub_position, u32 index, u32 out[2]) {
    out[0] = kb::to_monty((type << 26) + ((index >> 16) << 10) + sub_position);
    out[1] = kb::to_monty(index & 0xFFFF);
}
}  // namespace

struct lmh_type1_witness {
    u32 public_input[8];
    std::vector<u32> input_data, pubkeys;
    std::vector<u64> name_entry_begin, entry_offset;
    std::unique_ptr<u32[]> data;
    u64 n_sigs = 0;
    lm_vm_witness c;
    // late mode (lmh_aggregate_type_1): hash_pubkeys — the long chain — and what hangs on it (the digest inside `input_data`, the public
    // input) are computed by `hasher` while the VM already runs; `late` tells the runner which words are not final yet (lmh::VmLate)
    std::thread hasher;
    lmh::VmLate late;
    void finish() {
        if (hasher.joinable()) hasher.join();
    }
    ~lmh_type1_witness() { finish(); }
};

extern "C" {

static int type1_witness_build(const lmh_bytecode* bc, const uint32_t* raw_xmss, uint64_t n_raw, const uint32_t message[8], uint32_t slot,
                               bool late_mode, lmh_type1_witness** out);
int lmh_aggregate_type_1_witness(const lmh_bytecode* bc, const uint32_t* raw_xmss, uint64_t n_raw, const uint32_t message[8], uint32_t slot,
                                 lmh_type1_witness** out) {
    return type1_witness_build(bc, raw_xmss, n_raw, message, slot, false, out);
}
static int type1_witness_build(const lmh_bytecode* bc, const uint32_t* raw_xmss, uint64_t n_raw, const uint32_t message[8], uint32_t slot,
                               bool late_mode, lmh_type1_witness** out) {
    if (!bc || !raw_xmss || !message || !out || n_raw == 0) {
        lm_set_error("lmh_aggregate_type_1_witness: bad arguments (at least one signature)");
        return LM_E_INVALID;
    }
    *out = nullptr;
    const u32 n_names = lmh_bytecode_n_hint_names(bc);
    if (n_names && lmh_bytecode_hint_name_id(bc, "input_data") < 0) {
        lm_set_error("lmh_aggregate_type_1_witness: the bytecode carries no hint names (lmh_bytecode_set_hint_names) or is not the aggregation program");
        return LM_E_INVALID;
    }
    try {
        std::unique_ptr<lmh_type1_witness> w(new lmh_type1_witness());
        const bool times = getenv("LM_INPUT_TIMES") != nullptr;
        double tm[6] = {now_ms(), 0, 0, 0, 0, 0};
        // raw_xmss.sort_by(pk); dedup_by(pk) (:232-233): Ord of XmssPublicKey = merkle_root then public_param, canonical values
        std::vector<u32> key(n_raw * PUB_KEY_FLAT_SIZE);
        for (u64 i = 0; i < n_raw; i++)
            for (u32 k = 0; k < PUB_KEY_FLAT_SIZE; k++) key[i * 8 + k] = kb::from_monty(raw_xmss[i * LM_XMSS_SIG_WORDS + k]);
        std::vector<u32> order(n_raw);
        std::iota(order.begin(), order.end(), 0u);
        const u32* kp = key.data();
        std::stable_sort(order.begin(), order.end(), [kp](u32 a, u32 b) { return std::lexicographical_compare(kp + 8 * a, kp + 8 * a + 8, kp + 8 * b, kp + 8 * b + 8); });
        u64 n = 0;
        for (u64 i = 0; i < n_raw; i++)
            if (n == 0 || memcmp(kp + 8 * order[n - 1], kp + 8 * order[i], 32) != 0) order[n++] = order[i];
        if (n > MAX_XMSS_AGGREGATED) {
            lm_set_error("lmh_aggregate_type_1_witness: %llu signatures > MAX_XMSS_AGGREGATED", (unsigned long long)n);
            return LM_E_INVALID;
        }
        w->n_sigs = n;
        tm[1] = now_ms();
        // ---- the hint map (:317-365), flattened in name-id order: layout first, so that the streams can be filled while the hashes run -----
        const u32 log_size = lmh_bytecode_log_size(bc), n_vars = log_size + N_INSTRUCTION_COLUMNS_LOG;
        const u32 claim_size = (n_vars + 1) * DIMENSION, claim_padded = (claim_size + DIGEST_LEN - 1) / DIGEST_LEN * DIGEST_LEN;
        const u64 d_size = DIGEST_LEN + claim_padded + DIGEST_LEN + 4 * DIGEST_LEN;
        enum { S_NUM_CHUNKS, S_INPUT_DATA, S_META, S_PUBKEYS, S_RAW_INDICES, S_IS_SPLIT, S_WOTS, S_MERKLE, S_AGG_SIZES, S_TWEAKS, N_STREAMS };
        struct Stream {
            const char* name;
            u64 n_entries, words_per_entry;
            u32* dst;
        };
        Stream streams[N_STREAMS] = {{"input_data_num_chunks", 1, 1, nullptr}, {"input_data", 1, d_size, nullptr}, {"meta", 1, 3, nullptr},
                                     {"pubkeys", 1, n * PUB_KEY_FLAT_SIZE, nullptr}, {"raw_indices", 1, n, nullptr}, {"is_split", 1, 1, nullptr},
                                     {"wots", n, WOTS_SIG_SIZE_FE, nullptr}, {"xmss_merkle_node", n * LOG_LIFETIME, XMSS_DIGEST_LEN, nullptr},
                                     {"aggregate_sizes", 1, 0, nullptr}, {"tweak_table", 1, TWEAK_TABLE_SIZE_FE_PADDED, nullptr}};
        std::vector<int> stream_of(n_names, -1);
        u64 n_entries = 0, n_words = 0;
        for (int s = 0; s < N_STREAMS; s++) {
            const int id = lmh_bytecode_hint_name_id(bc, streams[s].name);
            if (id < 0) continue;  // (a stream the program never reads)
            stream_of[id] = s;
            n_entries += streams[s].n_entries, n_words += streams[s].n_entries * streams[s].words_per_entry;
        }
        w->name_entry_begin.assign(n_names + 1, 0);
        w->entry_offset.resize(n_entries + 1);
        w->data.reset(new u32[n_words ? n_words : 1]);
        u64 e = 0, o = 0;
        for (u32 id = 0; id < n_names; id++) {
            w->name_entry_begin[id] = e;
            const int s = stream_of[id];
            if (s < 0) continue;
            Stream& st = streams[s];
            for (u64 k = 0; k < st.n_entries; k++) w->entry_offset[e + k] = o + k * st.words_per_entry;
            st.dst = w->data.get() + o;
            e += st.n_entries, o += st.n_entries * st.words_per_entry;
        }
        // ---- a helper thread flattens the signatures and hashes the tweak table while this one hashes the public keys (the long chain) ----
        std::vector<u32> tw(TWEAK_TABLE_SIZE_FE_PADDED, 0u);
        u32 pubkeys_hash[8], tweaks_hash[8];
        const u32* ord = order.data();
        const bool threaded = n >= 64;  // (a thread costs ~30 us: not for a handful of signatures)
        const bool late = late_mode && threaded;  // hash_pubkeys runs beside the VM: its digest is a late word of the run (below)
        const bool main_copies_wots = late;       // (this thread has nothing to hash then: it takes the largest copy)
        auto copy_wots = [&]() {
            if (u32* dst = streams[S_WOTS].dst)  // encode_wots_signature (:188-194): randomness | chain_tips
                for (u64 i = 0; i < n; i++) memcpy(dst + i * WOTS_SIG_SIZE_FE, raw_xmss + (u64)ord[i] * LM_XMSS_SIG_WORDS + PUB_KEY_FLAT_SIZE, WOTS_SIG_SIZE_FE * 4);
        };
        auto side = [&]() {
            // compute_tweak_table(slot) (:124-151) and its hash (TWEAKS_HASHING_USE_IV = false)
            u32 at = 0;
            make_tweak(TWEAK_TYPE_ENCODING, 0, slot, &tw[at]), at += TWEAK_SLOT_SIZE;
            for (u32 i = 0; i < V * CHAIN_LENGTH; i++) make_tweak(TWEAK_TYPE_CHAIN, i, slot, &tw[at]), at += TWEAK_SLOT_SIZE;
            make_tweak(TWEAK_TYPE_WOTS_PK, 0, slot, &tw[at]), at += TWEAK_SLOT_SIZE;
            for (u32 level = 0; level < LOG_LIFETIME; level++)
                make_tweak(TWEAK_TYPE_MERKLE, level + 1, (u32)((u64)slot >> (level + 1)), &tw[at]), at += TWEAK_SLOT_SIZE;
            compress_slice(tw.data(), tw.size(), false, tweaks_hash);
            if (u32* dst = streams[S_TWEAKS].dst) memcpy(dst, tw.data(), tw.size() * 4);
            if (!main_copies_wots) copy_wots();
            if (u32* dst = streams[S_MERKLE].dst)
                for (u64 i = 0; i < n; i++)
                    memcpy(dst + i * LOG_LIFETIME * XMSS_DIGEST_LEN, raw_xmss + (u64)ord[i] * LM_XMSS_SIG_WORDS + PUB_KEY_FLAT_SIZE + WOTS_SIG_SIZE_FE,
                           LOG_LIFETIME * XMSS_DIGEST_LEN * 4);
            if (u32* dst = streams[S_RAW_INDICES].dst)
                for (u64 i = 0; i < n; i++) dst[i] = kb::to_monty((u32)i);  // global_pub_keys.binary_search(pk): the raw keys ARE the global list
        };
        tm[2] = now_ms();
        std::thread helper;
        if (threaded)
            helper = std::thread(side);
        else
            side();
        // global_pub_keys (= the raw keys: no children) and hash_pubkeys
        w->pubkeys.resize(n * PUB_KEY_FLAT_SIZE);
        for (u64 i = 0; i < n; i++) memcpy(&w->pubkeys[i * 8], raw_xmss + (u64)order[i] * LM_XMSS_SIG_WORDS, 32);
        if (late) {
            memset(pubkeys_hash, 0, sizeof pubkeys_hash);
            copy_wots();
        } else
            compress_slice(w->pubkeys.data(), w->pubkeys.size(), true, pubkeys_hash);
        if (u32* dst = streams[S_PUBKEYS].dst) memcpy(dst, w->pubkeys.data(), w->pubkeys.size() * 4);
        tm[3] = now_ms();
        if (threaded) helper.join();
        tm[4] = now_ms();
        // build_type1_input_data (:163-186) with the bytecode claim of a run without children: (0^n_vars, bytecode[0]) (bytecode_claims.rs:38-45)
        std::vector<u32>& d = w->input_data;
        d.assign(d_size, 0u);
        d[0] = kb::to_monty(TYPE1_FLAG), d[1] = kb::to_monty((u32)n);
        d[DIGEST_LEN + n_vars * DIMENSION] = lmh_bytecode_multilinear(bc)[0];  // EF::from(instructions_multilinear[0])
        u32 at = DIGEST_LEN + claim_padded;
        {
            alignas(64) u32 st[16];
            lmh_bytecode_hash(bc, st);
            for (int k = 0; k < 8; k++) st[8 + k] = kb::to_monty(SNARK_DOMAIN_SEP[k]);
            host_compress(st);  // poseidon16_compress_pair(bytecode_hash, SNARK_DOMAIN_SEP)
            memcpy(&d[at], st, 32), at += 8;
        }
        const u32 at_pubkeys_hash = at;
        memcpy(&d[at], pubkeys_hash, 32), at += 8;
        memcpy(&d[at], message, 32), at += 8;
        for (u32 c = 0; c < N_MERKLE_CHUNKS_FOR_SLOT; c++) d[at + c] = kb::to_monty((~(slot >> (4 * c))) & 0xF);
        at += 8;
        memcpy(&d[at], tweaks_hash, 32);
        if (!late) compress_slice(d.data(), d.size(), true, w->public_input);
        if (u32* dst = streams[S_NUM_CHUNKS].dst) dst[0] = kb::to_monty((u32)(d.size() / DIGEST_LEN));
        if (u32* dst = streams[S_INPUT_DATA].dst) memcpy(dst, d.data(), d.size() * 4);
        if (late) {
            // hash_pubkeys, its place in the input data (buffer and hint stream), the public input: on a thread of their own; the runner
            // is told which words it must not read yet (everything else of the witness is final when this function returns)
            lmh_type1_witness* wp = w.get();
            u32* hint_words = streams[S_INPUT_DATA].dst ? streams[S_INPUT_DATA].dst + at_pubkeys_hash : nullptr;
            memset(wp->public_input, 0, sizeof wp->public_input);
            wp->hasher = std::thread([wp, hint_words, at_pubkeys_hash]() {
                u32 h[8];
                compress_slice(wp->pubkeys.data(), wp->pubkeys.size(), true, h);
                memcpy(&wp->input_data[at_pubkeys_hash], h, 32);
                if (hint_words) memcpy(hint_words, h, 32);
                compress_slice(wp->input_data.data(), wp->input_data.size(), true, wp->public_input);
            });
            wp->late.n_ranges = hint_words ? 1 : 0;
            wp->late.first_word[0] = hint_words ? (u64)(hint_words - wp->data.get()) : 0;
            wp->late.n_words[0] = 8;
            wp->late.public_input = true;
            wp->late.wait = [wp]() { wp->finish(); };
        }
        if (u32* dst = streams[S_META].dst) dst[0] = 0, dst[1] = 0, dst[2] = kb::to_monty((u32)n);  // [n_recursions, n_dup, raw_count]
        if (u32* dst = streams[S_IS_SPLIT].dst) dst[0] = 0;
        w->name_entry_begin[n_names] = e;
        w->entry_offset[e] = o;
        w->c.preamble_memory_len = PREAMBLE_MEMORY_LEN, w->c.n_names = n_names;
        w->c.name_entry_begin = w->name_entry_begin.data(), w->c.entry_offset = w->entry_offset.data(), w->c.data = w->data.get();
        if (times)
            fprintf(stderr, "# aggregate_type_1 inputs: sort + dedup %.3f ms, hint map layout %.3f, hash_pubkeys (helper: tweak table, blobs) %.3f, join %.3f, input data + hash %.3f\n",
                    tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], now_ms() - tm[4]);
        *out = w.release();
    } catch (const std::bad_alloc&) {
        lm_set_error("lmh_aggregate_type_1_witness: out of memory");
        return LM_E_NOMEM;
    }
    return LM_OK;
}
const lm_vm_witness* lmh_type1_witness_vm(const lmh_type1_witness* w) { return &w->c; }
const uint32_t* lmh_type1_witness_public_input(const lmh_type1_witness* w) { return w->public_input; }
const uint32_t* lmh_type1_witness_input_data(const lmh_type1_witness* w, uint64_t* n_words) {
    if (n_words) *n_words = w->input_data.size();
    return w->input_data.data();
}
uint64_t lmh_type1_witness_n_sigs(const lmh_type1_witness* w) { return w->n_sigs; }
const uint32_t* lmh_type1_witness_pubkeys(const lmh_type1_witness* w) { return w->pubkeys.data(); }
void lmh_type1_witness_free(lmh_type1_witness* w) { delete w; }

int lmh_aggregate_type_1(lm_ctx* ctx, lmh_prover* p, const lmh_bytecode* bc, const uint32_t* raw_xmss, uint64_t n_raw, const uint32_t message[8],
                         uint32_t slot, const lm_whir_builder* builder, uint32_t n_threads, double times_ms[4], lm_vm_run_info* info) {
    const double t0 = now_ms();
    lmh_type1_witness* w = nullptr;
    // LM_INPUTS_EAGER=1: hash_pubkeys before the VM starts (the reference's order; A/B measurements)
    static const bool late_ok = getenv("LM_INPUTS_EAGER") == nullptr;
    int rc = type1_witness_build(bc, raw_xmss, n_raw, message, slot, late_ok, &w);
    if (rc) return rc;
    const double t1 = now_ms();
    double t[3] = {0, 0, 0};
    if (w->late.wait) lmh::vm_set_late(&w->late);  // (consumed by the run that follows on this thread)
    rc = lmh_prove_execution_vm_info(ctx, p, bc, w->public_input, 8, &w->c, builder, n_threads, t, info);
    lmh::vm_set_late(nullptr);
    w->finish();
    lmh_type1_witness_free(w);
    if (times_ms) times_ms[0] = t1 - t0, times_ms[1] = t[0], times_ms[2] = t[1], times_ms[3] = t[2];
    return rc;
}

}  // extern "C"
#endif
