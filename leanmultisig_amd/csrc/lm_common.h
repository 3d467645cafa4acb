// Shared internals of the HIP library (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>
#include <map>
#include <mutex>
#include <utility>
#include "../../include/leanmultisig.h"
#include "kb.h"
#include "poseidon16.h"

using kb::EF;
using kb::u32;
using kb::u64;

// generator of the 2^24-th roots of unity, canonical value (reference koala_bear.rs:50-54, last entry);
// every smaller two-adic generator is a repeated square of it.
static constexpr u32 LM_G24_CANON = 0x6ac49f88u;
static constexpr int LM_TW_LOG = 24;        // big table: w_{2^24}^j for j < 2^23   (32 MiB of HBM)
static constexpr int LM_TW_SMALL_LOG = 13;  // layered table: entry (1 << q) + j = w_{2^(q+1)}^j, q < 13

void lm_set_error(const char* fmt, ...);

#define LM_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            lm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return LM_E_DEVICE;                                                           \
        }                                                                                 \
    } while (0)

#define LM_REQUIRE(cond)                                                   \
    do {                                                                   \
        if (!(cond)) {                                                     \
            lm_set_error("%s:%d requirement failed: %s", __FILE__, __LINE__, #cond); \
            return LM_E_INVALID;                                           \
        }                                                                  \
    } while (0)

struct lm_ctx {
    unsigned long long uid = 0;  // process-unique: objects that outlive a context (a device-resident lmh_execution) find out through lm_ctx_by_uid
    int device = 0;
    hipStream_t stream = nullptr;
    u32* d_tw = nullptr;        // 2^(LM_TW_LOG-1) words
    u32* d_tw_small = nullptr;  // 2^LM_TW_SMALL_LOG words
    u32* d_sync = nullptr;      // [0] PoW result (0xffffffff when idle), [1] "writers done" counter of multi-block publishers
    unsigned long long* d_acc = nullptr;  // LM_ACC_WORDS accumulators of lm_grid_sum (zero between kernels)
    u32* d_coop = nullptr;      // COOP_TAB_WORDS: per-lane coefficient table of the 16-lane Poseidon (poseidon16_coop.h)
    u32* d_quad = nullptr;      // QUAD_TAB_WORDS: per-class coefficient table of the 4-lane Poseidon (poseidon16_quad.h)
    u32* d_scratch = nullptr;   // small reusable scratch (partials, points)
    u64 scratch_words = 0;
    // pinned, device-visible result buffer: final reduction kernels store round results here directly, the host
    // reads them after a stream synchronise (no D2H copy command per sumcheck round)
    u32* h_res = nullptr;
    static constexpr u64 RES_WORDS = 8192;  // (the resident GKR tail: 1024 + 64 slots of 64 words)
    static constexpr u64 RES_FLAG = RES_WORDS;  // one extra word after the payload: sequence number of the last result
    static constexpr u64 ERR_WORD = 8;          // h_res[RES_FLAG + ERR_WORD]: sticky count of data errors seen by kernels (lm_access_errors)
    u32 res_seq = 0;                            // host side counter; a publishing kernel stores it to h_res[RES_FLAG]
    // deferred results (lm_results_defer_begin / _end): evaluations that no transcript step separates are enqueued back to back, each
    // publishing at the next free offset of h_res; _end waits for the last sequence number and hands every result to its caller's buffer
    struct Deferred {
        u32* out;
        u32 offset, words;
    };
    bool defer_on = false;
    u32 defer_off = 0, defer_seq = 0;
    std::vector<Deferred> deferred;
    // pinned, device-visible mailbox lines (host -> resident kernel): line 0 belongs to `stream`, line 1 + k to aux_stream[k].
    // A resident kernel (k_gkr_tail) polls its line for the next message instead of ending and being relaunched.
    u32* h_cmd = nullptr;
    static constexpr u64 CMD_LINE_WORDS = 16;
    // Late-bound challenges ("launch ahead", lm_mail_*): a kernel whose only unknown arguments are the next two challenges is enqueued
    // BEHIND the kernel whose result they are derived from and waits for them on the device — line MAIL_LINE of h_cmd, relayed to the
    // other workgroups through d_relay — so the exchange costs the two posted writes instead of a launch (~13 -> ~7 us).  Messages are
    // numbered 1, 2, ..: consecutive numbers alternate the parity tag of the payload words, a kernel is told its number at launch.
    static constexpr u64 MAIL_LINE = 1 + 3;   // (= 1 + N_AUX: behind the lines of the resident GKR tails)
    static constexpr u64 CMD_LINES = 2 + 3;
    u32* d_relay = nullptr;                   // CMD_LINE_WORDS device words
    u32 mail_reserved = 0, mail_posted = 0;   // numbers handed to enqueued kernels / written to the line
    // pinned staging ring for small host -> device tables (pointer lists, job lists, evaluation points): the host image is
    // written here and copied with ONE asynchronous command — no synchronisation to keep a caller's vector alive, no
    // pageable-memory staging inside the runtime.  A region stays valid until the ring wraps; wrapping synchronises.
    uint8_t* h_stage = nullptr;
    static constexpr size_t STAGE_BYTES = 8u << 20;
    size_t stage_off = 0;
    // side streams: independent chains of one protocol step (the AIR sessions of a batched sumcheck round) run concurrently,
    // each publishing into its own flag word h_res[RES_FLAG + 1 + k].  Forked from / joined to `stream` with events.
    static constexpr int N_AUX = 3;
    hipStream_t aux_stream[N_AUX] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_event = nullptr;
    // caching device allocator: freed blocks are kept per size class and reused (hipMalloc/hipFree synchronise the
    // device; a proof performs ~100 allocations).  Single stream => reuse is stream-ordered and safe.
    std::mutex pool_mu;  // lm_pool_alloc / lm_pool_free may be reached from another thread (a device-resident lmh_execution released by a garbage collector)
    std::multimap<u64, void*> pool_free;
    std::map<void*, u64> pool_size;   // every block this pool owns -> its size class
    std::map<void*, bool> pool_in_use;  // handed out and not yet freed (a second lm_pool_free of the same pointer is ignored)
    u64 pool_bytes = 0;
    // optional per-kernel HIP-event timing (bench.py roofline leg): only launches whose kernel name is selected
    std::string prof_select;    // empty = profiling off; "*" = every kernel
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof_events;
    hipEvent_t prof_origin = nullptr;  // recorded by lm_profile_select: the time origin of lm_profile_busy_ms
    double prof_last_busy_ms = 0.0;
    std::map<std::string, u64> prof_bytes;  // algorithmic bytes of the recorded launches of a kernel (LM_PROF_BYTES at its launch sites)
    // device copies of long-lived host objects (a bytecode's instruction table, decoded records, hints), keyed by the object's
    // process-unique id: pool allocations of THIS context, gone with it (lm_ctx_cache_get / _put) — a cache inside the host object
    // keyed by the context's address would hand a dangling pointer to the next context allocated at the same address
    std::map<unsigned long long, void*> object_cache;
    // host <-> device exchanges (lm_wait_log): how long each lm_wait_result of the prover thread waited, in microseconds
    bool wait_log_on = false;
    std::vector<float> wait_us;
    // GKR layers that were re-run with one launch per exchange because a resident kernel never got its wave slots (lm_gkr_round)
    u32 soft_fallbacks = 0;
    // after a fallback the context keeps off resident kernels for a while (2 s, doubled by every further fallback up to 64 s, back to
    // 2 s after a clean period): a device that starved one tail will starve the next, and each failed attempt costs its 3 s timeout
    double no_resident_until_s = 0.0, no_resident_backoff_s = 0.0;
};

#define LM_LAUNCH(ctx, kernel, grid, block, shmem, ...) LM_LAUNCH_ON(ctx, (ctx)->stream, kernel, grid, block, shmem, __VA_ARGS__)
#define LM_LAUNCH_ON(ctx, strm, kernel, grid, block, shmem, ...)                                         \
    do {                                                                                                 \
        lm_ctx* c__ = (ctx);                                                                             \
        hipStream_t s__ = (strm);                                                                        \
        const bool p__ = !c__->prof_select.empty() &&                                                    \
                         (c__->prof_select == "*" || lm_prof_match(c__->prof_select.c_str(), #kernel));  \
        hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                       \
        if (p__) {                                                                                       \
            (void)hipEventCreate(&e0__);                                                                 \
            (void)hipEventCreate(&e1__);                                                                 \
            (void)hipEventRecord(e0__, s__);                                                     \
        }                                                                                                \
        hipLaunchKernelGGL(kernel, grid, block, shmem, s__, __VA_ARGS__);                        \
        if (p__) {                                                                                       \
            (void)hipEventRecord(e1__, s__);                                                     \
            c__->prof_events[#kernel].emplace_back(e0__, e1__);                                          \
        }                                                                                                \
    } while (0)

// Publish / wait protocol for small per-round results: the last kernel of a round writes its values into the pinned
// host-visible buffer, fences at system scope and stores the round's sequence number; the host spins on that word
// instead of paying a stream-synchronise per sumcheck round (~300 rounds per proof).
#if defined(__HIPCC__)
// Cross-block hand-over inside one kernel WITHOUT release / acquire fences.  On this multi-XCD part an agent-scope release
// fence is `buffer_wbl2`: it writes back every dirty line of the XCD's L2 — in a kernel that is streaming hundreds of MB of
// outputs that costs ~70 ns per block (k_fold_round with 4096 blocks: 504 -> 850 us).  The few words that actually cross
// blocks are instead written with agent-scope relaxed atomic stores (write-through, sc1), the wave waits until they are
// acknowledged (s_waitcnt vmcnt(0): what the memory model's release sequence does after its write-back), and the consumer
// reads them with agent-scope atomic loads (served past the non-coherent L2s).
// This hand-over is outside the HIP memory model: it relies on gfx9 behaviour (vmcnt counts stores and no-return atomics;
// sc1 write-through stores to fine-grained memory travel one posted-write path in order).  It is therefore tied to the targets
// it was validated on — gfx942 / gfx950 (on gfx10+ stores are counted by vscnt, which s_waitcnt(0) does not cover) — and can be
// switched off at build time: -DLM_PUBLISH_FENCES=1 restores release fences before every flag / ticket (the memory-model
// version: ~+4 ms per proof).  tools/stress_inflight.py checks every proof of concurrent provers with
// lmh_verify_execution, so a reordering would show up as a rejected proof.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LM_PUBLISH_FENCES) && !defined(__gfx950__) && !defined(__gfx942__)
#error "fence-free publish path validated on gfx942 / gfx950 only: build with -DLM_PUBLISH_FENCES=1 for other targets"
#endif
__device__ __forceinline__ void lm_store_agent(kb::u32* p, kb::u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ kb::u32 lm_load_agent(const kb::u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lm_store_system(kb::u32* p, kb::u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void lm_wait_stores() {
#if defined(LM_PUBLISH_FENCES)
    __threadfence_system();  // conservative build: a real release of everything this wave wrote
#else
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ kb::u32 lm_ticket(kb::u32* counter) {
#if defined(LM_PUBLISH_FENCES)
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Multi-block reduction without a second launch and without a pass over per-block partials: every block adds its N field
// words (< 2^31 each: 2^33 blocks fit) into 64-bit accumulators with agent-scope integer atomics, waits until they are
// performed, and takes a ticket; the block that draws the last ticket swaps the totals out (re-zeroing the accumulators for
// the next kernel on the stream) and reduces them mod p — the representation is additive, so the sum of Montgomery residues
// is the residue of the sum.  (Measured alternatives, profiles/r02_gridsum_notes.txt: per-block partials + a reducing pass by
// the last block costs ~7 us more per launch — three dependent trips to memory; a separate reducing kernel ~6 us + a launch.)
// `vals_lds` must be valid in threads < N of every block.  Returns true in ALL threads of the last block, totals in
// out_lds[0..N) (may alias vals_lds).
static constexpr kb::u32 LM_ACC_WORDS = 64;
template <int N>
__device__ __forceinline__ bool lm_grid_sum(const kb::u32* vals_lds, unsigned long long* acc, kb::u32* done_counter, kb::u32* out_lds) {
    static_assert(N <= (int)LM_ACC_WORDS, "accumulator count");
    __shared__ kb::u32 lm_is_last;
    if (threadIdx.x < N) {
        (void)__hip_atomic_fetch_add(acc + threadIdx.x, (unsigned long long)vals_lds[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lm_wait_stores();
    }
    __syncthreads();
    if (threadIdx.x == 0) lm_is_last = lm_ticket(done_counter) == gridDim.x - 1;
    __syncthreads();
    if (!lm_is_last) return false;
    if (threadIdx.x < N)
        out_lds[threadIdx.x] = (kb::u32)(__hip_atomic_exchange(acc + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % kb::P);
    if (threadIdx.x == 0) lm_store_agent(done_counter, 0);  // re-armed for the next kernel on the stream (kernel boundary orders it)
    __syncthreads();
    return true;
}
// Publishing is fence-free as well (a system-scope release is a write-back of the XCD's L2 plus an invalidate: microseconds
// when the previous kernel left it dirty, paid ~250 times per proof).  Contract: the payload is written with lm_store_system
// (write-through), every writing wave executes lm_wait_stores() (its stores are acknowledged) before the barrier that
// precedes this call, and ONE thread then stores the sequence number.  Payload and flag travel the same posted-write path.
__device__ __forceinline__ void lm_publish_flag_word(kb::u32* flag_word, kb::u32 seq) {
    lm_wait_stores();
#if defined(LM_PUBLISH_FENCES)
    __hip_atomic_store(flag_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    __hip_atomic_store(flag_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
__device__ __forceinline__ void lm_publish_flag(kb::u32* h_res, kb::u32 seq) { lm_publish_flag_word(h_res + lm_ctx::RES_FLAG, seq); }
#endif
#if defined(__HIPCC__)
static constexpr kb::u32 LM_MAIL_ABORT = 0xdead0002u;
static constexpr kb::u32 LM_MAIL_DISMISSED_BIT = 0x80000000u;
static constexpr unsigned long long LM_MAIL_TIMEOUT = 300000000ull;  // wall_clock64 ticks (100 MHz): 3 s without the message = abandoned
// Wait for message `no` (two challenges).  Every thread of the workgroup calls it; workgroup 0 polls the pinned line and relays the
// eleven tagged words to d_relay with agent-scope stores, the others poll the relay (a message is accepted when its ten payload words
// carry the parity of `no` and word 10 equals `no` in ONE wave-wide load: no ordering between the stores is assumed, a stale or
// half-written line is refused).  lds: 16 words.  Returns false when the kernel was dismissed (word 11) or nothing came for 3 s.
__device__ __forceinline__ bool lm_mail_receive(const kb::u32* __restrict__ h_line, kb::u32* __restrict__ d_relay, kb::u32 no,
                                                kb::u32* lds, kb::EF& r0, kb::EF& r1) {
    // A dismissal travels to the other workgroups in relay word 10 itself, as `no | LM_MAIL_DISMISSED_BIT` (message numbers stay below
    // 2^31): ONE word, tied to the number this kernel waits for — a dismissal left behind by an earlier kernel names another number and is
    // not honoured (round-5 advisor finding: a sticky abort word made every later launch leave without publishing).
    if (threadIdx.x < 64) {
        const kb::u32 lane = threadIdx.x;
        const bool first = blockIdx.x == 0;
        const unsigned long long t0 = wall_clock64();
        kb::u32 v = 0;
        bool done = false, dismissed = false;
        while (!done) {
            v = 0;
            if (lane < 16)
                v = first ? __hip_atomic_load(h_line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                          : __hip_atomic_load(d_relay + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane < 10 ? (v >> 31) == (no & 1) : lane == 10 ? v == no : true;
            done = __ballot(ok) == ~0ull;
            // (the clock is read by the scalar unit: the timeout decision is the same in every lane)
            const bool late = (unsigned long long)__builtin_amdgcn_readfirstlane((int)((wall_clock64() - t0) > LM_MAIL_TIMEOUT)) != 0;
            dismissed = __ballot(first ? (lane == 11 && v == LM_MAIL_ABORT) : (lane == 10 && v == (no | LM_MAIL_DISMISSED_BIT))) != 0 || late;
            if (dismissed) break;
            if (!done && !first) __builtin_amdgcn_s_sleep(4);
        }
        if (first) {
            if (dismissed) {
                if (lane == 10) __hip_atomic_store(d_relay + 10, no | LM_MAIL_DISMISSED_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (lane < 11) {
                __hip_atomic_store(d_relay + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane < 16) lds[lane] = dismissed ? LM_MAIL_ABORT : (v & 0x7fffffffu);
    }
    __syncthreads();
    if (lds[11] == LM_MAIL_ABORT) return false;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        r0.v[k] = __builtin_amdgcn_readfirstlane(lds[k]);
        r1.v[k] = __builtin_amdgcn_readfirstlane(lds[5 + k]);
    }
    return true;
}
#endif
// host side (lm_core.hip): the number the next enqueued kernel will wait for; the message; dismissal of kernels whose message
// will never come (error paths: synchronises the stream)
kb::u32 lm_mail_reserve(lm_ctx* ctx);
void lm_mail_post(lm_ctx* ctx, kb::u32 no, const kb::u32 r0[5], const kb::u32 r1[5]);
int lm_mail_abort(lm_ctx* ctx);
void lm_mail_reset(lm_ctx* ctx);  // line and relay back to "no message, no dismissal" whatever the counters say (the stream must be idle)
static inline const kb::u32* lm_mail_line(const lm_ctx* ctx) { return ctx->h_cmd + lm_ctx::CMD_LINE_WORDS * lm_ctx::MAIL_LINE; }
size_t lm_ctx_live_count();  // contexts alive in this process
int lm_gkr_foreign_processes(int device);  // live prover processes other than this one registered on the device (0 when there is no shared counter)
void lm_gkr_register_process(int device);  // lm_gkr.hip: take this process's slot in the device's shared counter (lm_ctx_create)
int lm_wait_result(lm_ctx* ctx, kb::u32 seq);
// the same on flag word h_res[RES_FLAG + 1 + aux] (aux >= 0), published by work on aux_stream[aux]
int lm_wait_result_aux(lm_ctx* ctx, int aux, kb::u32 seq);
int lm_aux_stream(lm_ctx* ctx, int aux, hipStream_t* out);  // created on first use

static inline bool lm_prof_match(const char* want, const char* name) {  // `want`: one kernel name or a comma-separated list
    if (name[0] == '(') name++;  // template kernels are launched as (k<...>)
    for (const char* w = want; *w;) {
        const char* e = strchr(w, ',');
        const size_t n = e ? (size_t)(e - w) : strlen(w);
        if (n && strncmp(w, name, n) == 0 && (name[n] == 0 || name[n] == '<')) return true;
        if (!e) break;
        w = e + 1;
    }
    return false;
}
// algorithmic HBM bytes of a launch that is being profiled (the roofline leg of bench.py divides them by the HIP-event time)
#define LM_PROF_BYTES(ctx, kernel, bytes)                                                                                    \
    do {                                                                                                                     \
        lm_ctx* cb__ = (ctx);                                                                                                \
        if (!cb__->prof_select.empty() && (cb__->prof_select == "*" || lm_prof_match(cb__->prof_select.c_str(), #kernel)))    \
            cb__->prof_bytes[#kernel] += (u64)(bytes);                                                                       \
    } while (0)

struct lm_tree {
    u32* d_matrix = nullptr;   // column-major: stored_cols x h words
    u32* d_digests = nullptr;  // all layers bottom-up, (2h - 1) x 8 words
    u32 log_h = 0;
    u32 is_ext = 0;
    u32 n_cols = 0;          // 2^folding_factor (in EF or base elements)
    u32 eff_cols = 0;        // effective (non-zero) columns, same unit
    u32 stored_words = 0;    // base columns stored per row (eff_cols or 5 * eff_cols)
    u32 leaf_words = 0;      // n_cols or 5 * n_cols
};

int lm_scratch(lm_ctx* ctx, u64 words, u32** out);
// bytes of the staging ring (nullptr in *out if the request is too large for it: the caller falls back to a synchronous copy)
int lm_stage_alloc(lm_ctx* ctx, size_t bytes, void** out);
// asynchronous host -> device copy of `bytes` BYTES on ctx->stream through the staging ring; `src` may be freed on return.
// (Not an overload of the public lm_upload(ctx, uint32_t*, const uint32_t*, n_words): with u32 pointers overload resolution
// would silently pick that one and read n_words WORDS.)
int lm_stage_upload(lm_ctx* ctx, void* d_dst, const void* src, size_t bytes);
// device -> host for a few words without a copy command or a stream synchronise: one tiny kernel stores them (and n1 more
// from a second source) into the pinned result buffer at `res_offset` and publishes; the host spins on the flag.
int lm_fetch_words(lm_ctx* ctx, int aux, const kb::u32* d_src0, kb::u32 n0, const kb::u32* d_src1, kb::u32 n1, kb::u32 res_offset, kb::u32* out);
// the same in two halves: several fetches on different streams are enqueued before the first one is awaited
int lm_fetch_words_begin(lm_ctx* ctx, int aux, const kb::u32* d_src0, kb::u32 n0, const kb::u32* d_src1, kb::u32 n1, kb::u32 res_offset, kb::u32* seq_out);
int lm_fetch_words_end(lm_ctx* ctx, int aux, kb::u32 seq, kb::u32 res_offset, kb::u32 n, kb::u32* out);
// pooled device memory (see lm_ctx::pool_free); lm_pool_alloc returns hipErrorOutOfMemory on failure
hipError_t lm_pool_alloc(lm_ctx* ctx, void** out, u64 bytes);
void lm_pool_free(lm_ctx* ctx, void* p);
template <class T>
static inline hipError_t lm_pool_alloc_t(lm_ctx* ctx, T** out, u64 bytes) {
    return lm_pool_alloc(ctx, reinterpret_cast<void**>(out), bytes);
}
