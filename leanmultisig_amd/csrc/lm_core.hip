// Context, memory helpers, twiddle tables, multilinear evaluation.
#include <sys/prctl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <time.h>
#include "lm_common.h"
#include "poseidon16_coop.h"
#include "poseidon16_quad.h"

using namespace kb;

// internal (C++ linkage, lm_host_internal.h): live contexts by unique id.  An object that holds device memory of a context and may be
// freed after it (a device-resident lmh_execution released by a garbage collector) asks lm_ctx_by_uid first: a destroyed context took
// its pool — and the object's buffers — with it.
static std::mutex g_ctx_mu;
static std::map<unsigned long long, lm_ctx*> g_ctx_live;
static unsigned long long g_ctx_next_uid = 1;
unsigned long long lm_ctx_uid(lm_ctx* ctx) { return ctx->uid; }
lm_ctx* lm_ctx_by_uid(unsigned long long uid) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx_live.find(uid);
    return it == g_ctx_live.end() ? nullptr : it->second;
}
// fn(ctx) under the registry lock when the context with that id is still alive (lm_ctx_destroy waits in ctx_unregister): a release
// that may run on any thread cannot race with the destruction of the context whose pool it returns blocks to
bool lm_ctx_with_live(unsigned long long uid, lm_ctx* expect, void (*fn)(lm_ctx*, void*), void* arg) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx_live.find(uid);
    if (it == g_ctx_live.end() || it->second != expect) return false;
    fn(expect, arg);
    return true;
}
static void ctx_register(lm_ctx* c) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    c->uid = g_ctx_next_uid++;
    g_ctx_live[c->uid] = c;
}
// contexts alive in this process (launch-ahead, lm_gkr.hip, is for a prover that has the device to itself)
size_t lm_ctx_live_count() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    return g_ctx_live.size();
}
static void ctx_unregister(lm_ctx* c) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctx_live.erase(c->uid);
}
// internal (C++ linkage, lm_host_internal.h): per-context cache of device copies of long-lived host objects
void* lm_ctx_cache_get(lm_ctx* ctx, unsigned long long key) {
    auto it = ctx->object_cache.find(key);
    return it == ctx->object_cache.end() ? nullptr : it->second;
}
void lm_ctx_cache_put(lm_ctx* ctx, unsigned long long key, void* p) { ctx->object_cache[key] = p; }
static thread_local char g_err[512] = "";
void lm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// tw_big[j] = w_{2^24}^j ; tw_small[(1 << q) + j] = w_{2^(q+1)}^j
__global__ void k_init_twiddles(u32* tw_big, u32* tw_small, u32 g24) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (1ull << (LM_TW_LOG - 1))) tw_big[i] = kb::pow(g24, i);
    if (i >= 1 && i < (1ull << LM_TW_SMALL_LOG)) {
        u32 q = 31 - __clz((u32)i);
        u64 j = i - (1ull << q);
        tw_small[i] = kb::pow(g24, j << (LM_TW_LOG - (q + 1)));
    }
    if (i == 0) tw_small[0] = ONE;
}

__global__ void k_aos_to_soa(const u32* __restrict__ aos, u32* __restrict__ soa, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 5; k++) soa[(u64)k * n + i] = aos[i * 5 + k];
}
__global__ void k_soa_to_aos(const u32* __restrict__ soa, u32* __restrict__ aos, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 5; k++) aos[i * 5 + k] = soa[(u64)k * n + i];
}

// ---------------------------------------------------------------------------------------------------------
// eq table (SoA, n_out = 2^n entries): out[i] = prod_j (i_j p_j + (1 - i_j)(1 - p_j)), point[0] <-> MSB of i.
// One entry per thread: n EF multiplications.  Only used for small tables (<= 2^16).
// point: device, n x 5 words AoS.
// ---------------------------------------------------------------------------------------------------------
// the coordinates travel as a kernel argument (no host-to-device copy per evaluation)
struct EqSmallArg {
    u32 v[20 * 5];
};
__global__ void k_eq_table_small(EqSmallArg point, u32 n, u32* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 len = 1u << n;
    if (i >= len) return;
    EF acc = ef_one();
    for (u32 j = 0; j < n; j++) {
        EF p;
#pragma unroll
        for (int k = 0; k < 5; k++) p.v[k] = point.v[j * 5 + k];
        u32 bit = (i >> (n - 1 - j)) & 1;
        EF f = bit ? p : ef_sub(ef_one(), p);
        acc = ef_mul(acc, f);
    }
#pragma unroll
    for (int k = 0; k < 5; k++) out[(u64)k * len + i] = acc.v[k];
}

// up to four such tables in one launch (the hi and lo tables of an evaluation, or of two evaluation points): table t takes the blocks
// [first_block[t], first_block[t + 1]) and is written at out + off[t]
struct EqMultiArg {
    EqSmallArg point[4];
    u32 n[4], first_block[5];
    u64 off[4];
};
__global__ __launch_bounds__(256) void k_eq_table_multi(EqMultiArg a, u32 n_tables, u32* __restrict__ out) {
    u32 t = 0;
    while (t + 1 < n_tables && blockIdx.x >= a.first_block[t + 1]) t++;
    const u32 n = a.n[t], len = 1u << n;
    const u32 i = (blockIdx.x - a.first_block[t]) * 256 + threadIdx.x;
    if (i >= len) return;
    EF acc = ef_one();
    for (u32 j = 0; j < n; j++) {
        EF p;
#pragma unroll
        for (int k = 0; k < 5; k++) p.v[k] = a.point[t].v[j * 5 + k];
        const u32 bit = (i >> (n - 1 - j)) & 1;
        acc = ef_mul(acc, bit ? p : ef_sub(ef_one(), p));
    }
    u32* o = out + a.off[t];
#pragma unroll
    for (int k = 0; k < 5; k++) o[(u64)k * len + i] = acc.v[k];
}

__device__ __forceinline__ EF wave_reduce_ef(EF v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        EF o;
#pragma unroll
        for (int k = 0; k < 5; k++) o.v[k] = __shfl_down(v.v[k], off, 64);
        v = ef_add(v, o);
    }
    return v;
}
// result valid in thread 0
__device__ __forceinline__ EF block_reduce_ef(EF v, u32* lds /* >= 20 words */) {
    v = wave_reduce_ef(v);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) lds[wave * 5 + k] = v.v[k];
    }
    __syncthreads();
    EF r = ef_zero();
    if (threadIdx.x == 0) {
        for (u32 w = 0; w < (blockDim.x >> 6); w++) {
            EF o;
#pragma unroll
            for (int k = 0; k < 5; k++) o.v[k] = lds[w * 5 + k];
            r = ef_add(r, o);
        }
    }
    return r;
}

// partial[(poly * n_hi + hi)] = eq_hi[hi] * sum_lo v[hi * 2^k_lo + lo] * eq_lo[lo]
// base values: 5 u64 accumulators with a fold every product (acc < 2^32 p, product < p^2, sum < 2^64).
// A workgroup evaluates MLE_POLYS polynomials on its slice: every eq_lo value it loads (5 words per element, from L2) is used
// for all of them — with one polynomial per workgroup the kernel moved 5 table words per data word and ran at the L2's pace
// (2.3 TB/s of HBM-equivalent traffic).
static constexpr u32 MLE_POLYS_MAX = 4;
template <u32 MLE_POLYS, class PolyAt>
__device__ __forceinline__ void mle_partial_base_impl(PolyAt poly_at, u32 n_polys, u32 k_lo, const u32* __restrict__ eq_lo,
                                                      const u32* __restrict__ eq_hi, u32 n_hi, u32* __restrict__ partial) {
    __shared__ u32 red[32];
    const u32 hi = blockIdx.x, poly0 = blockIdx.y * MLE_POLYS;
    const u32 len_lo = 1u << k_lo;
    const u32* v[MLE_POLYS];
#pragma unroll
    for (u32 q = 0; q < MLE_POLYS; q++) v[q] = poly_at(poly0 + q < n_polys ? poly0 + q : poly0) + (u64)hi * len_lo;
    u64 acc[MLE_POLYS][5];
#pragma unroll
    for (u32 q = 0; q < MLE_POLYS; q++)
#pragma unroll
        for (int k = 0; k < 5; k++) acc[q][k] = 0;
    for (u32 i = threadIdx.x; i < len_lo; i += 256) {
        u32 e[5], x[MLE_POLYS];
#pragma unroll
        for (int k = 0; k < 5; k++) e[k] = eq_lo[(u64)k * len_lo + i];
#pragma unroll
        for (u32 q = 0; q < MLE_POLYS; q++) x[q] = v[q][i];
#pragma unroll
        for (u32 q = 0; q < MLE_POLYS; q++)
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const u64 t = acc[q][k] + (u64)x[q] * e[k];
                const u64 y = t - P_SHL32;
                acc[q][k] = t >= P_SHL32 ? y : t;
            }
    }
    EF e;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) e.v[k] = eq_hi[(u64)k * n_hi + hi];
    }
#pragma unroll
    for (u32 q = 0; q < MLE_POLYS; q++) {
        EF s;
#pragma unroll
        for (int k = 0; k < 5; k++) s.v[k] = reduce(acc[q][k]);
        EF r = block_reduce_ef(s, red);
        if (threadIdx.x == 0 && poly0 + q < n_polys) {
            r = ef_mul(r, e);
#pragma unroll
            for (int k = 0; k < 5; k++) partial[((u64)(poly0 + q) * n_hi + hi) * 5 + k] = r.v[k];
        }
    }
}
template <u32 MLE_POLYS>
__global__ __launch_bounds__(256) void k_mle_partial_base(const u32* __restrict__ evals, u64 stride_words, u32 n_polys, u32 k_lo,
                                                          const u32* __restrict__ eq_lo, const u32* __restrict__ eq_hi,
                                                          u32 n_hi, u32* __restrict__ partial) {
    mle_partial_base_impl<MLE_POLYS>([&](u32 p) { return evals + (u64)p * stride_words; }, n_polys, k_lo, eq_lo, eq_hi, n_hi, partial);
}
// same with one device pointer per polynomial
template <u32 MLE_POLYS>
__global__ __launch_bounds__(256) void k_mle_partial_cols(const u32* const* __restrict__ cols, u32 n_polys, u32 k_lo,
                                                          const u32* __restrict__ eq_lo, const u32* __restrict__ eq_hi,
                                                          u32 n_hi, u32* __restrict__ partial) {
    mle_partial_base_impl<MLE_POLYS>([&](u32 p) { return cols[p]; }, n_polys, k_lo, eq_lo, eq_hi, n_hi, partial);
}
// same with one device pointer per polynomial
__global__ __launch_bounds__(256) void k_mle_partial_cols(const u32* const* __restrict__ cols, u32 k_lo,
                                                          const u32* __restrict__ eq_lo, const u32* __restrict__ eq_hi,
                                                          u32 n_hi, u32* __restrict__ partial) {
    __shared__ u32 red[32];
    const u32 hi = blockIdx.x, poly = blockIdx.y;
    const u32 len_lo = 1u << k_lo;
    const u32* v = cols[poly] + (u64)hi * len_lo;
    u64 acc[5] = {0, 0, 0, 0, 0};
    for (u32 i = threadIdx.x; i < len_lo; i += 256) {
        u32 x = v[i];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            u64 t = acc[k] + (u64)x * eq_lo[(u64)k * len_lo + i];
            u64 y = t - P_SHL32;
            acc[k] = t >= P_SHL32 ? y : t;
        }
    }
    EF s;
#pragma unroll
    for (int k = 0; k < 5; k++) s.v[k] = reduce(acc[k]);
    EF r = block_reduce_ef(s, red);
    if (threadIdx.x == 0) {
        EF e;
#pragma unroll
        for (int k = 0; k < 5; k++) e.v[k] = eq_hi[(u64)k * n_hi + hi];
        r = ef_mul(r, e);
#pragma unroll
        for (int k = 0; k < 5; k++) partial[((u64)poly * n_hi + hi) * 5 + k] = r.v[k];
    }
}
__global__ __launch_bounds__(256) void k_mle_partial_ext(const u32* __restrict__ evals, u64 stride_words, u64 plane,
                                                         u32 k_lo, const u32* __restrict__ eq_lo,
                                                         const u32* __restrict__ eq_hi, u32 n_hi,
                                                         u32* __restrict__ partial) {
    __shared__ u32 red[32];
    const u32 hi = blockIdx.x, poly = blockIdx.y;
    const u32 len_lo = 1u << k_lo;
    const u32* v = evals + (u64)poly * stride_words + (u64)hi * len_lo;
    EF s = ef_zero();
    for (u32 i = threadIdx.x; i < len_lo; i += 256) {
        EF x, e;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            x.v[k] = v[(u64)k * plane + i];
            e.v[k] = eq_lo[(u64)k * len_lo + i];
        }
        s = ef_add(s, ef_mul(x, e));
    }
    EF r = block_reduce_ef(s, red);
    if (threadIdx.x == 0) {
        EF e;
#pragma unroll
        for (int k = 0; k < 5; k++) e.v[k] = eq_hi[(u64)k * n_hi + hi];
        r = ef_mul(r, e);
#pragma unroll
        for (int k = 0; k < 5; k++) partial[((u64)poly * n_hi + hi) * 5 + k] = r.v[k];
    }
}
// ONE polynomial at NP points in one pass over its values (the OOD samples of a commitment are drawn together, whir/src/utils.rs:30-57):
// eq tables of point q at eq_lo + q * 5 * 2^k_lo and eq_hi + q * 5 * n_hi; partial[(q * n_hi + hi)] as above
template <u32 NP, bool EXT>
__global__ __launch_bounds__(256) void k_mle_partial_pts(const u32* __restrict__ evals, u64 plane, u32 k_lo, const u32* __restrict__ eq_lo,
                                                         const u32* __restrict__ eq_hi, u32 n_hi, u32* __restrict__ partial) {
    __shared__ u32 red[32];
    const u32 hi = blockIdx.x;
    const u32 len_lo = 1u << k_lo;
    const u32* v = evals + (u64)hi * len_lo;
    EF s[NP];
    if constexpr (EXT) {
#pragma unroll
        for (u32 q = 0; q < NP; q++) s[q] = ef_zero();
        for (u32 i = threadIdx.x; i < len_lo; i += 256) {
            EF x;
#pragma unroll
            for (int k = 0; k < 5; k++) x.v[k] = v[(u64)k * plane + i];
#pragma unroll
            for (u32 q = 0; q < NP; q++) {
                EF e;
#pragma unroll
                for (int k = 0; k < 5; k++) e.v[k] = eq_lo[((u64)q * 5 + k) * len_lo + i];
                s[q] = ef_add(s[q], ef_mul(x, e));
            }
        }
    } else {
        u64 acc[NP][5];
#pragma unroll
        for (u32 q = 0; q < NP; q++)
#pragma unroll
            for (int k = 0; k < 5; k++) acc[q][k] = 0;
        for (u32 i = threadIdx.x; i < len_lo; i += 256) {
            const u32 x = v[i];
#pragma unroll
            for (u32 q = 0; q < NP; q++)
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const u64 t = acc[q][k] + (u64)x * eq_lo[((u64)q * 5 + k) * len_lo + i];
                    const u64 y = t - P_SHL32;
                    acc[q][k] = t >= P_SHL32 ? y : t;
                }
        }
#pragma unroll
        for (u32 q = 0; q < NP; q++)
#pragma unroll
            for (int k = 0; k < 5; k++) s[q].v[k] = reduce(acc[q][k]);
    }
#pragma unroll
    for (u32 q = 0; q < NP; q++) {
        EF r = block_reduce_ef(s[q], red);
        if (threadIdx.x == 0) {
            EF e;
#pragma unroll
            for (int k = 0; k < 5; k++) e.v[k] = eq_hi[((u64)q * 5 + k) * n_hi + hi];
            r = ef_mul(r, e);
#pragma unroll
            for (int k = 0; k < 5; k++) partial[((u64)q * n_hi + hi) * 5 + k] = r.v[k];
        }
    }
}
// out[poly] = sum_hi partial[poly][hi]   (AoS EF)
// publish_to != NULL: out is the pinned result buffer; the last block to finish stores the sequence number
__global__ __launch_bounds__(256) void k_sum_partials(const u32* __restrict__ partial, u32 n_hi, u32* __restrict__ out,
                                                      u32* __restrict__ done_counter, u32* publish_to, u32 seq) {
    __shared__ u32 red[32];
    const u32 poly = blockIdx.x;
    EF s = ef_zero();
    for (u32 i = threadIdx.x; i < n_hi; i += 256) {
        EF x;
#pragma unroll
        for (int k = 0; k < 5; k++) x.v[k] = partial[((u64)poly * n_hi + i) * 5 + k];
        s = ef_add(s, x);
    }
    EF r = block_reduce_ef(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (publish_to)
                lm_store_system(out + poly * 5 + k, r.v[k]);
            else
                out[poly * 5 + k] = r.v[k];
        }
        if (publish_to) {
            lm_wait_stores();
            if (lm_ticket(done_counter) == gridDim.x - 1) {
                lm_store_agent(done_counter, 0);
                lm_publish_flag(publish_to, seq);
            }
        }
    }
}

static u64 pool_class(u64 bytes) {
    // 8 size classes per octave (<= 12.5 % slack), minimum 256 B
    if (bytes < 256) return 256;
    u64 hi = 1ull << (63 - __builtin_clzll(bytes));
    u64 step = hi >> 3;
    return (bytes + step - 1) / step * step;
}
hipError_t lm_pool_alloc(lm_ctx* ctx, void** out, u64 bytes) {
    const u64 cls = pool_class(bytes);
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    auto it = ctx->pool_free.find(cls);
    if (it != ctx->pool_free.end()) {
        *out = it->second;
        ctx->pool_in_use[it->second] = true;
        ctx->pool_free.erase(it);
        return hipSuccess;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) {
        // release cached blocks and retry once
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& kv : ctx->pool_free) {
            (void)hipFree(kv.second);
            ctx->pool_size.erase(kv.second);
            ctx->pool_in_use.erase(kv.second);
            ctx->pool_bytes -= kv.first;
        }
        ctx->pool_free.clear();
        e = hipMalloc(&p, cls);
        if (e != hipSuccess) return e;
    }
    ctx->pool_size[p] = cls;
    ctx->pool_in_use[p] = true;
    ctx->pool_bytes += cls;
    *out = p;
    return hipSuccess;
}
void lm_pool_free(lm_ctx* ctx, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    auto it = ctx->pool_size.find(p);
    if (it == ctx->pool_size.end()) {  // not ours
        (void)hipFree(p);
        return;
    }
    bool& in_use = ctx->pool_in_use[p];
    if (!in_use) return;  // double free: the block is already in the free list (inserting it twice would alias two later allocations)
    in_use = false;
    ctx->pool_free.emplace(it->second, p);
}

// Side streams are created on first use, not with the context: the runtime deals streams onto its few hardware queues
// round-robin in creation order, and a process that creates "main, side, side, side" per context puts every context's main
// stream on the same hardware queue (ten provers in one process: 110 k -> 78 k signatures/s).
int lm_aux_stream(lm_ctx* ctx, int aux, hipStream_t* out) {
    LM_REQUIRE(ctx && out && aux >= 0 && aux < lm_ctx::N_AUX);
    if (!ctx->aux_stream[aux]) LM_HIP(hipStreamCreateWithFlags(&ctx->aux_stream[aux], hipStreamNonBlocking));
    *out = ctx->aux_stream[aux];
    return LM_OK;
}
extern "C" int lm_wait_log(lm_ctx* ctx, int on) {
    LM_REQUIRE(ctx);
    ctx->wait_log_on = on != 0;
    ctx->wait_us.clear();
    return LM_OK;
}
extern "C" uint64_t lm_wait_log_read(lm_ctx* ctx, float* out_us, uint64_t cap) {
    if (!ctx) return 0;
    const uint64_t n = ctx->wait_us.size();
    if (out_us)
        for (uint64_t i = 0; i < n && i < cap; i++) out_us[i] = ctx->wait_us[i];
    return n;
}
int lm_wait_result(lm_ctx* ctx, u32 seq) { return lm_wait_result_aux(ctx, -1, seq); }
int lm_wait_result_aux(lm_ctx* ctx, int aux, u32 seq) {
    // While results are being deferred (lm_results_defer_begin) the head of the pinned buffer belongs to them: a call that publishes
    // there and waits — every round kernel, lm_mle_eval_points, lm_tree_open, ... — would overwrite the first deferred result.  The
    // header says "do not"; this makes a slip loud instead of a wrong transcript (lm_results_defer_end clears the flag before it waits).
    if (aux < 0 && ctx->defer_on) {
        lm_set_error("lm_wait_result: a result was published and awaited on the main flag while results are being deferred (lm_results_defer_begin .. _end)");
        return LM_E_INVALID;
    }
    volatile u32* flag = ctx->h_res + lm_ctx::RES_FLAG + (aux + 1);
    hipStream_t stream = aux < 0 ? ctx->stream : ctx->aux_stream[aux];
    // several publishers may be in flight on the stream (their sequence numbers increase): "at least seq" is the condition
    auto reached = [&] { return (int32_t)(*flag - seq) >= 0; };
    // Several provers in one process (leaves in flight): every prover thread spinning here is a CPU that the VM runs of the other
    // leaves do not get (the GPU boxes run under a quota of 16 CPUs).  When at least LM_WAIT_NAP_WAITERS threads (default 4) are
    // waiting at once, a wait that has already polled LM_WAIT_NAP times (default 3000, ~30 us) sleeps ~20 us between polls.  A
    // lone prover never sleeps: its latency is the metric.  LM_WAIT_NAP=0 switches the naps off.
    static const u64 nap_after = [] {
        const char* e = getenv("LM_WAIT_NAP");
        return e ? (u64)strtoull(e, nullptr, 10) : 3000ull;
    }();
    static const int nap_waiters = [] {
        const char* e = getenv("LM_WAIT_NAP_WAITERS");
        return e ? atoi(e) : 4;
    }();
    // Hybrid wait (LM_WAIT_MODE=hybrid, or auto = default: when another prover PROCESS is registered on this device): a rank of a sharded
    // job costs a core only while it has work.  The first ~20 us are polled as ever (most exchanges end there); a longer wait — a large
    // kernel, or a GPU shared with other ranks — sleeps ~10 us at a time (timer slack 1 us) between polls.  Eight ranks on sixteen CPUs
    // each spinning through 20 ms proofs starved each other's host work (host-busy per rank-step 5.6 -> 21-30 ms, profiles/r05_ranks_on_one_gpu.txt).
    // A lone prover keeps spinning: its latency is the metric (LM_WAIT_MODE=spin forces that everywhere).
    static const int wait_mode = [] {  // 0 spin, 1 hybrid, 2 auto
        const char* e = getenv("LM_WAIT_MODE");
        if (!e) return 2;
        return e[0] == 's' ? 0 : e[0] == 'h' ? 1 : 2;
    }();
    bool hybrid = wait_mode == 1;
    if (wait_mode == 2) {
        // (re-evaluated every 64 waits: another process may come or go; the check walks 64 words of shared memory)
        thread_local u32 tick = 0;
        thread_local bool shared_device = false;
        if ((tick++ & 63) == 0) shared_device = lm_gkr_foreign_processes(ctx->device) > 0;
        hybrid = shared_device;
    }
    static std::atomic<int> waiters{0};
    struct WaitScope {
        std::atomic<int>& w;
        explicit WaitScope(std::atomic<int>& x) : w(x) { w.fetch_add(1, std::memory_order_relaxed); }
        ~WaitScope() { w.fetch_sub(1, std::memory_order_relaxed); }
    } scope(waiters);
    struct WaitLog {  // lm_wait_log: the duration of this exchange as the prover thread saw it
        lm_ctx* c;
        std::chrono::steady_clock::time_point t0;
        explicit WaitLog(lm_ctx* x) : c(x) {
            if (c->wait_log_on) t0 = std::chrono::steady_clock::now();
        }
        ~WaitLog() {
            if (c->wait_log_on) c->wait_us.push_back(std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    } wait_log(ctx);
    for (u64 spins = 0; !reached(); spins++) {
        if (nap_after && spins >= nap_after && waiters.load(std::memory_order_relaxed) >= nap_waiters) {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
            spins += 2000;  // (a nap is worth ~2000 polls of wall clock: the give-up point below stays where it was)
        } else if (hybrid && spins >= 2000) {  // ~20 us of polling behind us
            thread_local bool slack_set = false;
            if (!slack_set) {
                (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);  // this thread's sleeps may end 1 us late, not 50
                slack_set = true;
            }
            struct timespec ts = {0, 8000};
            nanosleep(&ts, nullptr);
            spins += 1000;
        }
        if (spins > (1ull << 22)) {
            // Something is wrong or the kernel is long.  NOT hipStreamSynchronize: a resident kernel (k_gkr_tail) may own the stream
            // and be waiting for a mailbox message only this thread can write once it has seen the publication — a blocking sync
            // would return after the kernel's own timeout, with the flag set by a kernel that is gone.  Ask the runtime whether the
            // stream still has work and keep polling the flag; give up when the stream is idle without the publication, or after
            // LM_WAIT_LIMIT_S seconds (default 30) of wall clock.
            static const double limit_s = [] {
                const char* e = getenv("LM_WAIT_LIMIT_S");
                return e ? atof(e) : 30.0;
            }();
            const auto t_slow = std::chrono::steady_clock::now();
            for (u64 polls = 0; !reached(); polls++) {
                if ((polls & 1023) == 0) {
                    const hipError_t q = hipStreamQuery(stream);
                    if (q == hipSuccess) {
                        if (reached()) break;
                        lm_set_error("lm_wait_result: sequence %u never published (flag %u, stream idle)", seq, *flag);
                        return LM_E_DEVICE;
                    }
                    if (q != hipErrorNotReady) {
                        (void)hipGetLastError();
                        lm_set_error("lm_wait_result: %s while waiting for sequence %u", hipGetErrorString(q), seq);
                        return LM_E_DEVICE;
                    }
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_slow).count() > limit_s) {
                        lm_set_error("lm_wait_result: sequence %u not published after %.0f s (flag %u)", seq, limit_s, *flag);
                        return LM_E_DEVICE;
                    }
                    struct timespec ts = {0, 50000};
                    nanosleep(&ts, nullptr);
                }
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return LM_OK;
}

// ---- late-bound challenges (lm_common.h) ----
kb::u32 lm_mail_reserve(lm_ctx* ctx) { return ++ctx->mail_reserved; }
void lm_mail_post(lm_ctx* ctx, kb::u32 no, const kb::u32 r0[5], const kb::u32 r1[5]) {
    // (numbers are posted in the order they were reserved: the kernels wait in stream order)
    volatile u32* line = ctx->h_cmd + lm_ctx::CMD_LINE_WORDS * lm_ctx::MAIL_LINE;
    const u32 tag = (no & 1) << 31;
    for (int k = 0; k < 5; k++) line[k] = r0[k] | tag, line[5 + k] = r1[k] | tag;
    line[10] = no;
    ctx->mail_posted = no;
}
int lm_mail_abort(lm_ctx* ctx) {
    if (ctx->mail_posted == ctx->mail_reserved) return LM_OK;
    volatile u32* line = ctx->h_cmd + lm_ctx::CMD_LINE_WORDS * lm_ctx::MAIL_LINE;
    line[11] = LM_MAIL_ABORT;
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < lm_ctx::N_AUX; i++)  // (AIR sessions wait on their own streams)
        if (ctx->aux_stream[i]) (void)hipStreamSynchronize(ctx->aux_stream[i]);
    // the numbers skipped here never appear on the line: restart it so that the next message differs in parity from what is there
    const u32 last = ctx->mail_reserved, tag = (last & 1) << 31;
    u32 img[lm_ctx::CMD_LINE_WORDS];
    for (u32 i = 0; i < lm_ctx::CMD_LINE_WORDS; i++) img[i] = i < 10 ? tag : i == 10 ? last : 0;
    for (u32 i = 0; i < lm_ctx::CMD_LINE_WORDS; i++) line[i] = img[i];
    ctx->mail_posted = last;
    LM_HIP(hipMemcpy(ctx->d_relay, img, sizeof img, hipMemcpyHostToDevice));
    return LM_OK;
}

void lm_mail_reset(lm_ctx* ctx) {
    volatile u32* line = ctx->h_cmd + lm_ctx::CMD_LINE_WORDS * lm_ctx::MAIL_LINE;
    const u32 last = ctx->mail_reserved, tag = (last & 1) << 31;
    u32 img[lm_ctx::CMD_LINE_WORDS];
    for (u32 i = 0; i < lm_ctx::CMD_LINE_WORDS; i++) img[i] = i < 10 ? tag : i == 10 ? last : 0;
    for (u32 i = 0; i < lm_ctx::CMD_LINE_WORDS; i++) line[i] = img[i];
    ctx->mail_posted = last;
    (void)hipMemcpy(ctx->d_relay, img, sizeof img, hipMemcpyHostToDevice);
}
extern "C" uint32_t lm_soft_fallbacks(const lm_ctx* ctx) { return ctx ? ctx->soft_fallbacks : 0; }

int lm_stage_alloc(lm_ctx* ctx, size_t bytes, void** out) {
    *out = nullptr;
    bytes = (bytes + 63) & ~(size_t)63;
    if (!ctx->h_stage || bytes > lm_ctx::STAGE_BYTES / 4) return LM_OK;
    if (ctx->stage_off + bytes > lm_ctx::STAGE_BYTES) {  // wrap: everything that still reads the ring must have finished
        LM_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < lm_ctx::N_AUX; i++)
            if (ctx->aux_stream[i]) LM_HIP(hipStreamSynchronize(ctx->aux_stream[i]));
        ctx->stage_off = 0;
    }
    *out = ctx->h_stage + ctx->stage_off;
    ctx->stage_off += bytes;
    return LM_OK;
}
int lm_stage_upload(lm_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    if (bytes == 0) return LM_OK;
    void* st;
    int rc = lm_stage_alloc(ctx, bytes, &st);
    if (rc) return rc;
    if (!st) {
        LM_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        LM_HIP(hipStreamSynchronize(ctx->stream));
        return LM_OK;
    }
    memcpy(st, src, bytes);
    LM_HIP(hipMemcpyAsync(d_dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    return LM_OK;
}
__global__ __launch_bounds__(256) void k_publish_words(const u32* __restrict__ src0, u32 n0, const u32* __restrict__ src1, u32 n1,
                                                       u32* __restrict__ dst, u32* __restrict__ flag_word, u32 seq) {
    for (u32 i = threadIdx.x; i < n0 + n1; i += 256) lm_store_system(dst + i, i < n0 ? src0[i] : src1[i - n0]);
    lm_wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) lm_publish_flag_word(flag_word, seq);
}
int lm_fetch_words_begin(lm_ctx* ctx, int aux, const u32* d_src0, u32 n0, const u32* d_src1, u32 n1, u32 res_offset, u32* seq_out) {
    LM_REQUIRE(ctx && d_src0 && seq_out && aux >= -1 && aux < lm_ctx::N_AUX && (u64)res_offset + n0 + n1 <= lm_ctx::RES_WORDS);
    hipStream_t stream = ctx->stream;
    if (aux >= 0) {
        int rc = lm_aux_stream(ctx, aux, &stream);
        if (rc) return rc;
    }
    const u32 seq = ++ctx->res_seq;
    LM_LAUNCH_ON(ctx, stream, k_publish_words, dim3(1), dim3(256), 0, d_src0, n0, d_src1, n1, ctx->h_res + res_offset,
                 ctx->h_res + lm_ctx::RES_FLAG + 1 + aux, seq);
    LM_HIP(hipGetLastError());
    *seq_out = seq;
    return LM_OK;
}
int lm_fetch_words_end(lm_ctx* ctx, int aux, u32 seq, u32 res_offset, u32 n, u32* out) {
    LM_REQUIRE(ctx && out && (u64)res_offset + n <= lm_ctx::RES_WORDS);
    int rc = lm_wait_result_aux(ctx, aux, seq);
    if (rc) return rc;
    memcpy(out, ctx->h_res + res_offset, (size_t)n * 4);
    return LM_OK;
}
int lm_fetch_words(lm_ctx* ctx, int aux, const u32* d_src0, u32 n0, const u32* d_src1, u32 n1, u32 res_offset, u32* out) {
    u32 seq;
    int rc = lm_fetch_words_begin(ctx, aux, d_src0, n0, d_src1, n1, res_offset, &seq);
    return rc ? rc : lm_fetch_words_end(ctx, aux, seq, res_offset, n0 + n1, out);
}

int lm_scratch(lm_ctx* ctx, u64 words, u32** out) {
    if (words > ctx->scratch_words) {
        LM_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) LM_HIP(hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_words = 0;
        u64 w = words < (1ull << 20) ? (1ull << 20) : words;
        if (hipMalloc(&ctx->d_scratch, w * 4) != hipSuccess) {
            lm_set_error("scratch hipMalloc of %llu bytes failed", (unsigned long long)w * 4);
            return LM_E_NOMEM;
        }
        ctx->scratch_words = w;
    }
    *out = ctx->d_scratch;
    return LM_OK;
}

// stack_polynomials (crates/sub_protocols/src/stacked_pcs.rs:99-157): dst = zero-padded concatenation of column slices.
// One pass over the destination: every uint4 finds the job that covers it (jobs sorted by destination), or is zero.
struct StackJob {
    const u32* src;
    u64 begin, end;  // destination range in uint4 units
};
// Each workgroup copies CONTIGUOUS chunks of the destination (grid-stride over chunks of STACK_CHUNK vectors): a lane finds
// its job by binary search once per chunk and walks forward from there — the jobs are sorted and thousands of vectors long, so
// the walk almost never moves (a search per vector was ~8 dependent loads in front of every 16-byte copy).
static constexpr u64 STACK_CHUNK = 256 * 8;
__global__ __launch_bounds__(256) void k_stack_columns(uint4* __restrict__ dst, u64 n_vec, const StackJob* __restrict__ jobs,
                                                       u32 n_jobs) {
    const u64 n_chunks = (n_vec + STACK_CHUNK - 1) / STACK_CHUNK;
    for (u64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const u64 first = c * STACK_CHUNK + threadIdx.x;
        u32 lo = 0;
        if (n_jobs) {
            u32 hi = n_jobs - 1;  // last job with begin <= first
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (jobs[mid].begin <= first)
                    lo = mid;
                else
                    hi = mid - 1;
            }
        }
        StackJob j = n_jobs ? jobs[lo] : StackJob{nullptr, 0, 0};
#pragma unroll
        for (int u = 0; u < (int)(STACK_CHUNK / 256); u++) {
            const u64 e = first + (u64)u * 256;
            if (e >= n_vec) break;
            while (lo + 1 < n_jobs && jobs[lo + 1].begin <= e) j = jobs[++lo];
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n_jobs && j.begin <= e && e < j.end) v = reinterpret_cast<const uint4*>(j.src)[e - j.begin];
            dst[e] = v;
        }
    }
}

// col[c][i] = val[c] for i < count: the padding rows of every column of a table in ONE launch (a fill command per column — 160 of
// them per trace — kept the stream busy for 0.5 ms with ~3 us kernels 7 us apart)
struct FillCols {
    static constexpr u32 MAX = 128;
    u32* col[MAX];
    u32 val[MAX];
};
__global__ __launch_bounds__(256) void k_fill_columns(const FillCols f, u64 count) {
    u32* dst = f.col[blockIdx.y];
    const u32 v = f.val[blockIdx.y];
    for (u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += (u64)gridDim.x * 1024) {
        if (i + 4 <= count && ((uintptr_t)(dst + i) & 15) == 0)
            *reinterpret_cast<uint4*>(dst + i) = make_uint4(v, v, v, v);
        else
            for (u64 k = i; k < count && k < i + 4; k++) dst[k] = v;
    }
}

// which lane's value does lane 0 receive from a DPP row rotation by one? (defines the table of poseidon16_coop.h)
__global__ void k_coop_probe(u32* out) { out[threadIdx.x] = coop_rot<1>(threadIdx.x); }

extern "C" {

const char* lm_last_error(void) { return g_err; }

static int ctx_create_impl(int device, lm_ctx* c);
int lm_ctx_create(int device, lm_ctx** out) {
    LM_REQUIRE(out);
    LM_HIP(hipSetDevice(device));
    lm_ctx* c = new lm_ctx();
    ctx_register(c);
    const int rc = ctx_create_impl(device, c);
    if (rc) {
        lm_ctx_destroy(c);  // releases whatever was allocated before the failure
        return rc;
    }
    *out = c;
    return LM_OK;
}
static int ctx_create_impl(int device, lm_ctx* c) {
    c->device = device;
    LM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    LM_HIP(hipMalloc(&c->d_tw, (1ull << (LM_TW_LOG - 1)) * 4));
    LM_HIP(hipMalloc(&c->d_tw_small, (1ull << LM_TW_SMALL_LOG) * 4));
    LM_HIP(hipHostMalloc((void**)&c->h_res, (lm_ctx::RES_WORDS + 16) * 4, hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < 16; i++) c->h_res[lm_ctx::RES_FLAG + i] = 0;
    static_assert(lm_ctx::MAIL_LINE == 1 + lm_ctx::N_AUX && lm_ctx::CMD_LINES == 2 + lm_ctx::N_AUX, "mailbox lines");
    LM_HIP(hipHostMalloc((void**)&c->h_cmd, lm_ctx::CMD_LINE_WORDS * lm_ctx::CMD_LINES * 4, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_cmd, 0, lm_ctx::CMD_LINE_WORDS * lm_ctx::CMD_LINES * 4);
    LM_HIP(hipMalloc(&c->d_relay, lm_ctx::CMD_LINE_WORDS * 4));
    LM_HIP(hipMemset(c->d_relay, 0, lm_ctx::CMD_LINE_WORDS * 4));
    LM_HIP(hipHostMalloc((void**)&c->h_stage, lm_ctx::STAGE_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
    LM_HIP(hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming));
    const u64 n = 1ull << (LM_TW_LOG - 1);
    LM_LAUNCH(c, k_init_twiddles, dim3((unsigned)(n / 256)), dim3(256), 0, c->d_tw, c->d_tw_small,
                       to_monty(LM_G24_CANON));
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(c->stream));
    {
        const u32 init[4] = {0xffffffffu, 0, 0, 0};
        LM_HIP(hipMalloc(&c->d_sync, sizeof init));
        LM_HIP(hipMemcpy(c->d_sync, init, sizeof init, hipMemcpyHostToDevice));
        LM_HIP(hipMalloc(&c->d_acc, LM_ACC_WORDS * 8));
        LM_HIP(hipMemset(c->d_acc, 0, LM_ACC_WORDS * 8));
    }
    {  // 16-lane Poseidon: probe the DPP rotation direction, build the coefficient table for it
        LM_HIP(hipMalloc(&c->d_coop, COOP_TAB_WORDS * 4));
        LM_LAUNCH(c, k_coop_probe, dim3(1), dim3(64), 0, c->d_coop);
        u32 probe[64];
        LM_HIP(hipMemcpyAsync(probe, c->d_coop, sizeof probe, hipMemcpyDeviceToHost, c->stream));
        LM_HIP(hipStreamSynchronize(c->stream));
        std::vector<u32> tab(COOP_TAB_WORDS);
        bool ok = lm_coop_table_build(probe[0], tab.data());
        for (u32 l = 0; l < 64 && ok; l++)  // the rotation must stay inside each 16-lane row
            ok = probe[l] == (l & ~15u) + ((probe[0] == 15 ? (l & 15) + 15 : (l & 15) + 1) & 15);
        if (!ok) {
            lm_set_error("lm_ctx_create: unexpected DPP row rotation (lane 0 reads lane %u)", probe[0]);
            return LM_E_DEVICE;
        }
        LM_HIP(hipMemcpyAsync(c->d_coop, tab.data(), COOP_TAB_WORDS * 4, hipMemcpyHostToDevice, c->stream));
        LM_HIP(hipStreamSynchronize(c->stream));
        std::vector<u32> qt(QUAD_TAB_WORDS);  // 4-lane Poseidon (quad_perm semantics are fixed: no probe)
        lm_quad_table_build(qt.data());
        LM_HIP(hipMalloc(&c->d_quad, QUAD_TAB_WORDS * 4));
        LM_HIP(hipMemcpy(c->d_quad, qt.data(), QUAD_TAB_WORDS * 4, hipMemcpyHostToDevice));
    }
    lm_gkr_register_process(device);  // this process is on the device from now on (launch-ahead gate of the other provers, lm_gkr.hip)
    return LM_OK;
}
void lm_ctx_destroy(lm_ctx* c) {
    if (!c) return;
    ctx_unregister(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < lm_ctx::N_AUX; i++)
        if (c->aux_stream[i]) {
            (void)hipStreamSynchronize(c->aux_stream[i]);
            (void)hipStreamDestroy(c->aux_stream[i]);
        }
    if (c->fork_event) (void)hipEventDestroy(c->fork_event);
    if (c->prof_origin) (void)hipEventDestroy(c->prof_origin);
    if (c->d_tw) (void)hipFree(c->d_tw);
    if (c->d_tw_small) (void)hipFree(c->d_tw_small);
    if (c->d_coop) (void)hipFree(c->d_coop);
    if (c->d_quad) (void)hipFree(c->d_quad);
    if (c->d_sync) (void)hipFree(c->d_sync);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_cmd) (void)hipHostFree(c->h_cmd);
    if (c->d_relay) (void)hipFree(c->d_relay);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto& kv : c->pool_size) (void)hipFree(kv.first);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
int lm_bind_thread(lm_ctx* ctx) {
    LM_REQUIRE(ctx);
    LM_HIP(hipSetDevice(ctx->device));
    return LM_OK;
}
// rows of lm_access_counts jobs whose address range fell outside the image since the last reset (the reference panics on
// those; the device skips and counts them).  Synchronises the stream.
uint32_t lm_access_errors(lm_ctx* ctx, int reset) {
    if (!ctx) return 0;
    (void)hipStreamSynchronize(ctx->stream);
    volatile u32* w = ctx->h_res + lm_ctx::RES_FLAG + lm_ctx::ERR_WORD;
    const u32 v = *w;
    if (reset) *w = 0;
    return v;
}
int lm_sync(lm_ctx* ctx) {
    LM_REQUIRE(ctx);
    LM_HIP(hipStreamSynchronize(ctx->stream));
    return LM_OK;
}
void* lm_ctx_stream(lm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int lm_profile_select(lm_ctx* ctx, const char* kernel_name) {
    LM_REQUIRE(ctx);
    ctx->prof_select = kernel_name ? kernel_name : "";
    if (!ctx->prof_select.empty()) {  // time origin for lm_profile_busy_ms: in front of every launch that will be recorded
        if (!ctx->prof_origin) LM_HIP(hipEventCreate(&ctx->prof_origin));
        LM_HIP(hipEventRecord(ctx->prof_origin, ctx->stream));
    }
    return LM_OK;
}
double lm_profile_busy_ms(lm_ctx* ctx) { return ctx ? ctx->prof_last_busy_ms : 0.0; }
// names of the kernels that have recorded launches, '\n'-separated (template arguments stripped); returns the length needed
uint64_t lm_profile_names(lm_ctx* ctx, char* buf, uint64_t cap) {
    if (!ctx) return 0;
    std::string all;
    std::string last;
    for (auto& kv : ctx->prof_events) {
        std::string k = kv.first;
        if (!k.empty() && k[0] == '(') k = k.substr(1);
        const size_t lt = k.find('<');
        if (lt != std::string::npos) k = k.substr(0, lt);
        if (k == last) continue;
        last = k;
        all += k;
        all += '\n';
    }
    if (buf && cap) {
        const size_t n = std::min<size_t>(all.size(), cap - 1);
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return all.size() + 1;
}
// algorithmic HBM bytes recorded for the launches of `kernel_name` since the last call (only kernels whose launch sites carry
// LM_PROF_BYTES: k_prod_round2, k_fold2_round); clears the counter
uint64_t lm_profile_read_bytes(lm_ctx* ctx, const char* kernel_name) {
    if (!ctx || !kernel_name) return 0;
    auto it = ctx->prof_bytes.find(kernel_name);
    if (it == ctx->prof_bytes.end()) return 0;
    const u64 b = it->second;
    ctx->prof_bytes.erase(it);
    return b;
}
int lm_profile_read(lm_ctx* ctx, const char* kernel_name, uint64_t* n_launches, double* total_ms) {
    LM_REQUIRE(ctx && kernel_name && n_launches && total_ms);
    LM_HIP(hipStreamSynchronize(ctx->stream));
    *n_launches = 0;
    *total_ms = 0.0;
    // template instantiations are recorded as "name<args>": a bare name matches all of them
    const std::string want = kernel_name;
    std::vector<std::pair<double, double>> spans;  // (start, end) of every launch, ms after lm_profile_select
    for (auto it = ctx->prof_events.begin(); it != ctx->prof_events.end();) {
        std::string key = it->first;
        if (!key.empty() && key[0] == '(') key = key.substr(1);
        const bool match = key == want || (key.size() > want.size() && key.compare(0, want.size(), want) == 0 && key[want.size()] == '<');
        if (!match) {
            ++it;
            continue;
        }
        for (auto& pr : it->second) {
            float ms = 0.f, t0 = 0.f;
            LM_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            *total_ms += ms;
            *n_launches += 1;
            if (ctx->prof_origin && hipEventElapsedTime(&t0, ctx->prof_origin, pr.first) == hipSuccess) spans.emplace_back((double)t0, (double)t0 + ms);
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        it = ctx->prof_events.erase(it);
    }
    (void)hipGetLastError();
    // launches of one family on several streams overlap (the AIR sessions): the time during which at least one of them ran
    std::sort(spans.begin(), spans.end());
    double busy = 0.0, lo = 0.0, hi = -1.0;
    for (auto& sp : spans) {
        if (hi < lo || sp.first > hi) {
            if (hi >= lo) busy += hi - lo;
            lo = sp.first, hi = sp.second;
        } else if (sp.second > hi)
            hi = sp.second;
    }
    if (hi >= lo) busy += hi - lo;
    ctx->prof_last_busy_ms = busy;
    return LM_OK;
}

int lm_malloc(lm_ctx* ctx, uint64_t n_words, uint32_t** d_out) {
    LM_REQUIRE(ctx && d_out && n_words > 0);
    if (lm_pool_alloc_t(ctx, d_out, n_words * 4) != hipSuccess) {
        lm_set_error("device allocation of %llu bytes failed", (unsigned long long)n_words * 4);
        return LM_E_NOMEM;
    }
    return LM_OK;
}
int lm_free(lm_ctx* ctx, uint32_t* d_ptr) {
    LM_REQUIRE(ctx);
    lm_pool_free(ctx, d_ptr);
    return LM_OK;
}
int lm_upload(lm_ctx* ctx, uint32_t* d_dst, const uint32_t* src, uint64_t n_words) {
    LM_REQUIRE(ctx && d_dst && src);
    LM_HIP(hipMemcpyAsync(d_dst, src, n_words * 4, hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    return LM_OK;
}
int lm_upload_async(lm_ctx* ctx, uint32_t* d_dst, const uint32_t* src, uint64_t n_words) {
    LM_REQUIRE(ctx && d_dst && src);
    LM_HIP(hipMemcpyAsync(d_dst, src, n_words * 4, hipMemcpyHostToDevice, ctx->stream));
    return LM_OK;
}
int lm_download(lm_ctx* ctx, uint32_t* dst, const uint32_t* d_src, uint64_t n_words) {
    LM_REQUIRE(ctx && dst && d_src);
    LM_HIP(hipMemcpyAsync(dst, d_src, n_words * 4, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    return LM_OK;
}
int lm_memset_zero(lm_ctx* ctx, uint32_t* d_dst, uint64_t n_words) {
    LM_REQUIRE(ctx && d_dst);
    LM_HIP(hipMemsetAsync(d_dst, 0, n_words * 4, ctx->stream));
    return LM_OK;
}
int lm_fill_columns(lm_ctx* ctx, uint32_t* const* d_cols, const uint32_t* values, uint32_t n_cols, uint64_t offset, uint64_t count) {
    LM_REQUIRE(ctx && d_cols && values);
    if (n_cols == 0 || count == 0) return LM_OK;
    for (u32 c0 = 0; c0 < n_cols; c0 += FillCols::MAX) {  // pointers and values travel as kernel arguments
        FillCols f;
        const u32 n = std::min<u32>(FillCols::MAX, n_cols - c0);
        for (u32 c = 0; c < n; c++) {
            LM_REQUIRE(d_cols[c0 + c]);
            f.col[c] = d_cols[c0 + c] + offset;
            f.val[c] = values[c0 + c];
        }
        const unsigned bx = (unsigned)std::min<u64>((count + 1023) / 1024, 64);
        LM_LAUNCH(ctx, k_fill_columns, dim3(bx, n), dim3(256), 0, f, count);
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int lm_ef_aos_to_soa(lm_ctx* ctx, const uint32_t* d_aos, uint32_t* d_soa, uint64_t n) {
    LM_REQUIRE(ctx && d_aos && d_soa);
    if (!n) return LM_OK;
    LM_LAUNCH(ctx, k_aos_to_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_aos, d_soa, n);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int lm_ef_soa_to_aos(lm_ctx* ctx, const uint32_t* d_soa, uint32_t* d_aos, uint64_t n) {
    LM_REQUIRE(ctx && d_aos && d_soa);
    if (!n) return LM_OK;
    LM_LAUNCH(ctx, k_soa_to_aos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_soa, d_aos, n);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// offset of the next result in the pinned buffer: 0 unless results are being deferred (then the slot is recorded for _end; a result
// that no longer fits ends the deferral for this call: the earlier ones are collected first)
// (a failed collection of the earlier results fails this call too: their callers' buffers were not filled — round-5 advisor finding)
static int defer_slot(lm_ctx* ctx, u32* out, u32 words, u32* at_out) {
    *at_out = 0;
    if (!ctx->defer_on) return LM_OK;
    if (ctx->defer_off + words > lm_ctx::RES_WORDS) {
        const int rc = lm_results_defer_end(ctx);
        if (rc) return rc;  // (deferral is off now: the caller's lm_results_defer_end is a no-op)
        ctx->defer_on = true;
    }
    const u32 at = ctx->defer_off;
    ctx->deferred.push_back({out, at, words});
    ctx->defer_off += words;
    *at_out = at;
    return LM_OK;
}
int lm_results_defer_begin(lm_ctx* ctx) {
    LM_REQUIRE(ctx && !ctx->defer_on);
    ctx->defer_on = true;
    ctx->defer_off = 0;
    ctx->defer_seq = 0;
    ctx->deferred.clear();
    return LM_OK;
}
int lm_results_defer_end(lm_ctx* ctx) {
    LM_REQUIRE(ctx);
    if (!ctx->defer_on) return LM_OK;
    ctx->defer_on = false;
    int rc = ctx->defer_seq ? lm_wait_result(ctx, ctx->defer_seq) : LM_OK;
    if (!rc)
        for (const lm_ctx::Deferred& d : ctx->deferred) memcpy(d.out, ctx->h_res + d.offset, (size_t)d.words * 4);
    ctx->deferred.clear();
    ctx->defer_off = 0;
    ctx->defer_seq = 0;
    return rc;
}

// hi and lo eq tables of up to two points in ONE launch: table (q, hi) at base + q * 5 * n_hi (hi block) resp. the lo block
static int launch_eq_tables(lm_ctx* ctx, const uint32_t* points, u32 n_points, u32 n_vars, u32 k_hi, u32 k_lo, u32* d_eq_lo, u32* d_eq_hi,
                            u32* base) {
    EqMultiArg a;
    memset(&a, 0, sizeof a);
    const u32 n_hi = 1u << k_hi, len_lo = 1u << k_lo;
    u32 t = 0, blocks = 0;
    for (u32 q = 0; q < n_points; q++) {
        const u32* pt = points + (size_t)q * n_vars * 5;
        if (k_hi) memcpy(a.point[t].v, pt, (size_t)k_hi * 20);
        a.n[t] = k_hi, a.first_block[t] = blocks, a.off[t] = (u64)(d_eq_hi - base) + (u64)q * 5 * n_hi;
        blocks += (n_hi + 255) / 256, t++;
        if (k_lo) memcpy(a.point[t].v, pt + (size_t)k_hi * 5, (size_t)k_lo * 20);
        a.n[t] = k_lo, a.first_block[t] = blocks, a.off[t] = (u64)(d_eq_lo - base) + (u64)q * 5 * len_lo;
        blocks += (len_lo + 255) / 256, t++;
    }
    a.first_block[t] = blocks;
    LM_LAUNCH(ctx, k_eq_table_multi, dim3(blocks), dim3(256), 0, a, t, base);
    return LM_OK;
}

int lm_mle_eval_points(lm_ctx* ctx, const uint32_t* d_evals, int is_ext, uint32_t n_vars, uint32_t n_points, const uint32_t* points,
                       uint32_t* out) {
    LM_REQUIRE(ctx && d_evals && out && n_points >= 1 && n_vars <= 32 && (n_vars == 0 || points));
    const u32 k_lo = n_vars < 12 ? n_vars : 12;
    const u32 k_hi = n_vars - k_lo;
    LM_REQUIRE(k_hi <= 20);
    const u32 n_hi = 1u << k_hi, len_lo = 1u << k_lo;
    for (u32 q0 = 0; q0 < n_points; q0 += 2) {  // two points per pass over the polynomial
        const u32 np = n_points - q0 < 2 ? n_points - q0 : 2;
        const u64 need = 2 * (5ull * len_lo + 5ull * n_hi) + 2ull * n_hi * 5 + 64;
        u32* s;
        int rc = lm_scratch(ctx, need, &s);
        if (rc) return rc;
        u32* d_eq_lo = s;
        u32* d_eq_hi = d_eq_lo + 2 * 5ull * len_lo;
        u32* d_partial = d_eq_hi + 2 * 5ull * n_hi;
        if ((rc = launch_eq_tables(ctx, points + (size_t)q0 * n_vars * 5, np, n_vars, k_hi, k_lo, d_eq_lo, d_eq_hi, s))) return rc;
        const u64 plane = 1ull << n_vars;
        if (np == 2) {
            if (is_ext)
                LM_LAUNCH(ctx, (k_mle_partial_pts<2, true>), dim3(n_hi), dim3(256), 0, d_evals, plane, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
            else
                LM_LAUNCH(ctx, (k_mle_partial_pts<2, false>), dim3(n_hi), dim3(256), 0, d_evals, plane, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
        } else {
            if (is_ext)
                LM_LAUNCH(ctx, (k_mle_partial_pts<1, true>), dim3(n_hi), dim3(256), 0, d_evals, plane, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
            else
                LM_LAUNCH(ctx, (k_mle_partial_pts<1, false>), dim3(n_hi), dim3(256), 0, d_evals, plane, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
        }
        const u32 seq = ++ctx->res_seq;
        LM_LAUNCH(ctx, k_sum_partials, dim3(np), dim3(256), 0, (const u32*)d_partial, n_hi, ctx->h_res, ctx->d_sync + 1, ctx->h_res, seq);
        LM_HIP(hipGetLastError());
        if ((rc = lm_wait_result(ctx, seq))) return rc;
        memcpy(out + (size_t)q0 * 5, ctx->h_res, (size_t)np * 20);
    }
    return LM_OK;
}

int lm_mle_eval(lm_ctx* ctx, const uint32_t* d_evals, int is_ext, uint32_t n_vars, uint32_t n_polys,
                uint64_t stride_words, const uint32_t* point, uint32_t* out) {
    LM_REQUIRE(ctx && d_evals && out && n_polys >= 1 && n_vars <= 40);
    LM_REQUIRE(n_vars == 0 || point);
    const u32 k_lo = n_vars < 12 ? n_vars : 12;
    const u32 k_hi = n_vars - k_lo;
    LM_REQUIRE(k_hi <= 20);
    const u32 n_hi = 1u << k_hi, len_lo = 1u << k_lo;
    // scratch: point (n*5) | eq_lo (5*len_lo) | eq_hi (5*n_hi) | partial (n_polys*n_hi*5) | out (n_polys*5)
    const u64 need = (u64)n_vars * 5 + 5ull * len_lo + 5ull * n_hi + (u64)n_polys * n_hi * 5 + (u64)n_polys * 5 + 64;
    u32* s;
    int rc = lm_scratch(ctx, need, &s);
    if (rc) return rc;
    u32* d_eq_lo = s;
    u32* d_eq_hi = d_eq_lo + 5ull * len_lo;
    u32* d_partial = d_eq_hi + 5ull * n_hi;
    u32* d_out = d_partial + (u64)n_polys * n_hi * 5;
    // point = (hi part: first k_hi coordinates) ++ (lo part: last k_lo coordinates)
    if ((rc = launch_eq_tables(ctx, point, 1, n_vars, k_hi, k_lo, d_eq_lo, d_eq_hi, s))) return rc;
    if (!is_ext) {
        if (n_polys >= MLE_POLYS_MAX)
            LM_LAUNCH(ctx, k_mle_partial_base<MLE_POLYS_MAX>, dim3(n_hi, (n_polys + MLE_POLYS_MAX - 1) / MLE_POLYS_MAX), dim3(256), 0, d_evals,
                      stride_words, n_polys, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
        else
            LM_LAUNCH(ctx, k_mle_partial_base<1>, dim3(n_hi, n_polys), dim3(256), 0, d_evals, stride_words, n_polys, k_lo, d_eq_lo, d_eq_hi, n_hi,
                      d_partial);
    } else {
        LM_LAUNCH(ctx, k_mle_partial_ext, dim3(n_hi, n_polys), dim3(256), 0, d_evals, stride_words,
                           1ull << n_vars, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
    }
    const bool pinned = (u64)n_polys * 5 <= lm_ctx::RES_WORDS;
    if (pinned) {
        u32 at;
        if ((rc = defer_slot(ctx, out, n_polys * 5, &at))) return rc;
        const u32 seq = ++ctx->res_seq;
        LM_LAUNCH(ctx, k_sum_partials, dim3(n_polys), dim3(256), 0, (const u32*)d_partial, n_hi, ctx->h_res + at, ctx->d_sync + 1, ctx->h_res,
                  seq);
        LM_HIP(hipGetLastError());
        if (ctx->defer_on) {
            ctx->defer_seq = seq;
            return LM_OK;
        }
        if ((rc = lm_wait_result(ctx, seq))) return rc;
        memcpy(out, ctx->h_res, (u64)n_polys * 20);
    } else {
        LM_LAUNCH(ctx, k_sum_partials, dim3(n_polys), dim3(256), 0, (const u32*)d_partial, n_hi, d_out, ctx->d_sync + 1, (u32*)nullptr, 0u);
        LM_HIP(hipGetLastError());
        LM_HIP(hipMemcpyAsync(out, d_out, (u64)n_polys * 20, hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return LM_OK;
}


int lm_mle_eval_cols(lm_ctx* ctx, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t n_vars, const uint32_t* point,
                     uint32_t* out) {
    LM_REQUIRE(ctx && d_cols && out && n_cols >= 1 && n_vars <= 32 && (u64)n_cols * 5 <= lm_ctx::RES_WORDS);
    LM_REQUIRE(n_vars == 0 || point);
    const u32 k_lo = n_vars < 12 ? n_vars : 12;
    const u32 k_hi = n_vars - k_lo;
    const u32 n_hi = 1u << k_hi, len_lo = 1u << k_lo;
    LM_REQUIRE(k_hi <= 20);
    const u64 need = 2ull * n_cols + 5ull * len_lo + 5ull * n_hi + (u64)n_cols * n_hi * 5 + 96;
    u32* s;
    int rc = lm_scratch(ctx, need, &s);
    if (rc) return rc;
    const u32** d_ptrs = reinterpret_cast<const u32**>(s);
    u32* d_eq_lo = s + ((2ull * n_cols + 15) & ~15ull);
    u32* d_eq_hi = d_eq_lo + 5ull * len_lo;
    u32* d_partial = d_eq_hi + 5ull * n_hi;
    if ((rc = lm_stage_upload(ctx, (void*)d_ptrs, d_cols, (size_t)n_cols * 8))) return rc;
    if ((rc = launch_eq_tables(ctx, point, 1, n_vars, k_hi, k_lo, d_eq_lo, d_eq_hi, s))) return rc;
    if (n_cols >= MLE_POLYS_MAX)
        LM_LAUNCH(ctx, k_mle_partial_cols<MLE_POLYS_MAX>, dim3(n_hi, (n_cols + MLE_POLYS_MAX - 1) / MLE_POLYS_MAX), dim3(256), 0,
                  (const u32* const*)d_ptrs, n_cols, k_lo, d_eq_lo, d_eq_hi, n_hi, d_partial);
    else
        LM_LAUNCH(ctx, k_mle_partial_cols<1>, dim3(n_hi, n_cols), dim3(256), 0, (const u32* const*)d_ptrs, n_cols, k_lo, d_eq_lo, d_eq_hi, n_hi,
                  d_partial);
    u32 at;
    if ((rc = defer_slot(ctx, out, n_cols * 5, &at))) return rc;
    const u32 seq = ++ctx->res_seq;
    LM_LAUNCH(ctx, k_sum_partials, dim3(n_cols), dim3(256), 0, (const u32*)d_partial, n_hi, ctx->h_res + at, ctx->d_sync + 1, ctx->h_res, seq);
    LM_HIP(hipGetLastError());
    if (ctx->defer_on) {
        ctx->defer_seq = seq;
        return LM_OK;
    }
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(out, ctx->h_res, (u64)n_cols * 20);
    return LM_OK;
}
int lm_stack_columns(lm_ctx* ctx, uint32_t* d_dst, uint64_t total_words, uint32_t n_jobs, const uint32_t* const* d_src,
                     const uint64_t* dst_offset, const uint64_t* n_words) {
    LM_REQUIRE(ctx && d_dst && (total_words & 3) == 0 && ((uintptr_t)d_dst & 15) == 0);
    LM_REQUIRE(n_jobs == 0 || (d_src && dst_offset && n_words));
    if (total_words == 0) return LM_OK;
    bool vec_ok = true;
    u64 prev_end = 0;
    for (u32 i = 0; i < n_jobs; i++) {
        LM_REQUIRE(d_src[i] && dst_offset[i] >= prev_end && dst_offset[i] + n_words[i] <= total_words);
        prev_end = dst_offset[i] + n_words[i];
        if ((dst_offset[i] & 3) || (n_words[i] & 3) || ((uintptr_t)d_src[i] & 15)) vec_ok = false;
    }
    if (!vec_ok) {  // unaligned pieces: plain copies
        LM_HIP(hipMemsetAsync(d_dst, 0, total_words * 4, ctx->stream));
        for (u32 i = 0; i < n_jobs; i++)
            if (n_words[i]) LM_HIP(hipMemcpyAsync(d_dst + dst_offset[i], d_src[i], n_words[i] * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return LM_OK;
    }
    std::vector<StackJob> jobs;
    for (u32 i = 0; i < n_jobs; i++)
        if (n_words[i]) jobs.push_back({d_src[i], dst_offset[i] / 4, (dst_offset[i] + n_words[i]) / 4});
    u32* s;
    int rc = lm_scratch(ctx, (jobs.size() * sizeof(StackJob) + 3) / 4 + 16, &s);
    if (rc) return rc;
    if (!jobs.empty()) {
        if ((rc = lm_stage_upload(ctx, s, jobs.data(), jobs.size() * sizeof(StackJob)))) return rc;
    }
    const u64 n_vec = total_words / 4;
    const u32 blocks = (u32)std::min<u64>((n_vec + STACK_CHUNK - 1) / STACK_CHUNK, 16384);
    LM_LAUNCH(ctx, k_stack_columns, dim3(blocks), dim3(256), 0, reinterpret_cast<uint4*>(d_dst), n_vec, (const StackJob*)s,
              (u32)jobs.size());
    LM_HIP(hipGetLastError());
    return LM_OK;
}

int lm_copy_d2d(lm_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, uint64_t n_words) {
    LM_REQUIRE(ctx && d_dst && d_src);
    if (n_words) LM_HIP(hipMemcpyAsync(d_dst, d_src, n_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return LM_OK;
}

}  // extern "C"
