// GKR for a sum of fractions sum n_i / d_i  (reference: crates/sub_protocols/src/quotient_gkr/{mod,layers,sumcheck_utils}.rs).
//
// Device layout: natural index order (the reference's chunk-bit-reversed SIMD packing is a CPU artefact; all
// transcript values are layout independent).  A layer with 2^v entries is {nums, dens}: nums is one base plane for the
// input layer and SoA EF above it, dens is SoA EF.  Children of parent j are entries 2j (left) and 2j+1 (right), so one
// sumcheck pair (parents 2j', 2j'+1) is 4 consecutive entries: one 16-byte load per plane.
// During a layer's sumcheck the four multilinears (n_l, n_r, d_l, d_r) live as 4 SoA EF arrays that halve every round
// (LSB-first folding, sumcheck_utils.rs:278-357).
#include <algorithm>
#include "lm_common.h"
#include "lm_eqsplit.h"

using namespace kb;

struct lm_gkr {
    u32 n_vars = 0;
    const u32* d_nums0 = nullptr;  // caller's input layer (base)
    const u32* d_dens0 = nullptr;  // caller's input layer (SoA EF)
    std::vector<u32*> nums, dens;  // layers n_vars-1 .. 5 (index 0 = 2^(n_vars-1) entries), SoA EF, owned
    u32* work[2] = {nullptr, nullptr};  // ping-pong: 4 arrays x 5 planes
    u64 work_words = 0;
    PrefixEqTables eqt;  // prefix eq tables of the current layer
    // state of the layer being proven
    u32 K = 0;        // number of rounds = number of coordinates of the claim point
    u32 round = 0;
    int cur = -1;     // which work buffer holds the current arrays (-1: still in layer storage)
    u64 m = 0;        // current length of each of the 4 arrays
    EF alpha;
};

// ---- layer construction (layers.rs:124-189): (n0 d1 + n1 d0, d0 d1) ------------------------------------------------
template <bool BASE>
__global__ __launch_bounds__(256) void k_gkr_layer_up(const u32* __restrict__ n_in, const u32* __restrict__ d_in, u64 m_out,
                                                      u32* __restrict__ n_out, u32* __restrict__ d_out) {
    const u64 plane_in = 2 * m_out;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < m_out; i += (u64)gridDim.x * 256) {
        EF d0, d1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint2 v = *reinterpret_cast<const uint2*>(d_in + (u64)k * plane_in + 2 * i);
            d0.v[k] = v.x;
            d1.v[k] = v.y;
        }
        EF no;
        if (BASE) {
            uint2 v = *reinterpret_cast<const uint2*>(n_in + 2 * i);
            no = ef_add(ef_mul_base(d1, v.x), ef_mul_base(d0, v.y));
        } else {
            EF n0, n1;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                uint2 v = *reinterpret_cast<const uint2*>(n_in + (u64)k * plane_in + 2 * i);
                n0.v[k] = v.x;
                n1.v[k] = v.y;
            }
            no = ef_add(ef_mul(d1, n0), ef_mul(d0, n1));
        }
        EF dd = ef_mul(d0, d1);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            n_out[(u64)k * m_out + i] = no.v[k];
            d_out[(u64)k * m_out + i] = dd.v[k];
        }
    }
}

// pair_coeffs (sumcheck_utils.rs:65-79) accumulated with weight w into acc[0..4) = (c0_num, c2_num, c0_den, c2_den)
__device__ __forceinline__ void pair_accumulate(const EF& nl0, const EF& nl1, const EF& nr0, const EF& nr1, const EF& dl0,
                                                const EF& dl1, const EF& dr0, const EF& dr1, const EF& w, EF acc[4]) {
    const EF ddl = ef_sub(dl1, dl0), ddr = ef_sub(dr1, dr0);
    const EF c0d = ef_mul(dl0, dr0);
    const EF c2d = ef_mul(ddl, ddr);
    const EF c0n = ef_add(ef_mul(nl0, dr0), ef_mul(nr0, dl0));
    const EF c2n = ef_add(ef_mul(ef_sub(nl1, nl0), ddr), ef_mul(ef_sub(nr1, nr0), ddl));
    acc[0] = ef_add(acc[0], ef_mul(c0n, w));
    acc[1] = ef_add(acc[1], ef_mul(c2n, w));
    acc[2] = ef_add(acc[2], ef_mul(c0d, w));
    acc[3] = ef_add(acc[3], ef_mul(c2d, w));
}
__device__ __forceinline__ void pair_accumulate_base(u32 nl0, u32 nl1, u32 nr0, u32 nr1, const EF& dl0, const EF& dl1,
                                                     const EF& dr0, const EF& dr1, const EF& w, EF acc[4]) {
    const EF ddl = ef_sub(dl1, dl0), ddr = ef_sub(dr1, dr0);
    const EF c0d = ef_mul(dl0, dr0);
    const EF c2d = ef_mul(ddl, ddr);
    const EF c0n = ef_add(ef_mul_base(dr0, nl0), ef_mul_base(dl0, nr0));
    const EF c2n = ef_add(ef_mul_base(ddr, sub(nl1, nl0)), ef_mul_base(ddl, sub(nr1, nr0)));
    acc[0] = ef_add(acc[0], ef_mul(c0n, w));
    acc[1] = ef_add(acc[1], ef_mul(c2n, w));
    acc[2] = ef_add(acc[2], ef_mul(c0d, w));
    acc[3] = ef_add(acc[3], ef_mul(c2d, w));
}

// block-sum of 4 EF accumulators -> partial[block][20]
__device__ __forceinline__ void block_store_acc(const EF acc[4], u32* lds /* 80 words */, u32* dst) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 v[20];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int k = 0; k < 5; k++) v[a * 5 + k] = wave_sum_u32(acc[a].v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 20; k++) lds[wave * 20 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 20) {
        u32 s = 0;
        for (u32 w = 0; w < (blockDim.x >> 6); w++) s = add(s, lds[w * 20 + threadIdx.x]);
        dst[threadIdx.x] = s;
    }
}
// (c0_num + alpha c0_den, c2_num + alpha c2_den) from the 20 summed words
__device__ __forceinline__ void combine_alpha(const u32* tot, const EF& alpha, u32* out) {
    EF c0n, c2n, c0d, c2d;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        c0n.v[k] = tot[k];
        c2n.v[k] = tot[5 + k];
        c0d.v[k] = tot[10 + k];
        c2d.v[k] = tot[15 + k];
    }
    const EF a = ef_add(c0n, ef_mul(alpha, c0d)), b = ef_add(c2n, ef_mul(alpha, c2d));
#pragma unroll
    for (int k = 0; k < 5; k++) {
        out[k] = a.v[k];
        out[5 + k] = b.v[k];
    }
}
// End of a round kernel: a single-block launch finishes the round itself (result straight to the host-visible buffer),
// a multi-block launch leaves per-block partials for k_gkr_reduce.
__device__ __forceinline__ void finish_round(const EF acc[4], u32* lds, u32* tot, u32* partial, const EF& alpha, u32* final_out,
                                             u32 seq) {
    block_store_acc(acc, lds, tot);
    __syncthreads();
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) {
            combine_alpha(tot, alpha, final_out);
            lm_publish_flag(final_out, seq);
        }
    } else if (threadIdx.x < 20) {
        partial[(u64)blockIdx.x * 20 + threadIdx.x] = tot[threadIdx.x];
    }
}
// out[0..5) = c0_num + alpha c0_den ; out[5..10) = c2_num + alpha c2_den
__global__ __launch_bounds__(256) void k_gkr_reduce(const u32* __restrict__ partial, u32 n, EF alpha, u32* __restrict__ out,
                                                    u32 seq) {
    __shared__ u32 lds[80];
    EF acc[4] = {ef_zero(), ef_zero(), ef_zero(), ef_zero()};
    for (u32 i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int k = 0; k < 5; k++) acc[a].v[k] = add(acc[a].v[k], partial[(u64)i * 20 + a * 5 + k]);
    __shared__ u32 tot[20];
    block_store_acc(acc, lds, tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        combine_alpha(tot, alpha, out);
        lm_publish_flag(out, seq);
    }
}

// ---- round 0 of a layer straight from layer storage: pairs j' < n_pairs, entries 4j' .. 4j'+3 ------------------------
template <bool BASE>
__global__ __launch_bounds__(256) void k_gkr_round_storage(const u32* __restrict__ n_in, const u32* __restrict__ d_in,
                                                           u64 n_pairs, EqSplit eq, u32* __restrict__ partial, EF alpha,
                                                           u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 lds[80];
    __shared__ u32 tot[20];
    const u64 plane = 4 * n_pairs;
    EF acc[4] = {ef_zero(), ef_zero(), ef_zero(), ef_zero()};
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_pairs; j += (u64)gridDim.x * 256) {
        EF dl0, dr0, dl1, dr1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint4 v = *reinterpret_cast<const uint4*>(d_in + (u64)k * plane + 4 * j);
            dl0.v[k] = v.x;
            dr0.v[k] = v.y;
            dl1.v[k] = v.z;
            dr1.v[k] = v.w;
        }
        const EF w = eq_split_at(eq, j);
        if (BASE) {
            uint4 v = *reinterpret_cast<const uint4*>(n_in + 4 * j);
            pair_accumulate_base(v.x, v.z, v.y, v.w, dl0, dl1, dr0, dr1, w, acc);
        } else {
            EF nl0, nr0, nl1, nr1;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                uint4 v = *reinterpret_cast<const uint4*>(n_in + (u64)k * plane + 4 * j);
                nl0.v[k] = v.x;
                nr0.v[k] = v.y;
                nl1.v[k] = v.z;
                nr1.v[k] = v.w;
            }
            pair_accumulate(nl0, nl1, nr0, nr1, dl0, dl1, dr0, dr1, w, acc);
        }
    }
    finish_round(acc, lds, tot, partial, alpha, final_out, seq);
}

// ---- fold by r then compute the next round.  MODE 0: input = layer storage with base nums, 1: layer storage with EF
// nums, 2: four SoA arrays of length m_in.  Output: four SoA arrays of length m_out = m_in / 2 at `out`
// (array a at out + a * 5 * m_out).  Thread j' produces outputs 2j', 2j'+1 and (if m_out >= 2) their pair coefficients.
template <int MODE, bool LAST>
__global__ __launch_bounds__(256) void k_gkr_fold_round(const u32* __restrict__ n_in, const u32* __restrict__ d_in,
                                                        const u32* __restrict__ arr_in, u64 m_out, EF r, EqSplit eq,
                                                        u32* __restrict__ out, u32* __restrict__ partial, EF alpha,
                                                        u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 lds[80];
    __shared__ u32 tot[20];
    const u64 m_in = 2 * m_out;
    EF acc[4] = {ef_zero(), ef_zero(), ef_zero(), ef_zero()};
    // LAST: m_out == 1 (the final fold of a layer: one output per array, no pair to accumulate)
    constexpr int N_OUT = LAST ? 1 : 2;
    const u64 n_threads_work = LAST ? 1 : m_out / 2;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_threads_work; j += (u64)gridDim.x * 256) {
        EF o[N_OUT][4];  // [which output][array] — fully unrolled, stays in registers
#pragma unroll
        for (int t = 0; t < N_OUT; t++) {
            const u64 i = 2 * j + t;  // output index; inputs 2i, 2i+1 of each array
            EF a[4], b[4];
            if (MODE == 2) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        uint2 v = *reinterpret_cast<const uint2*>(arr_in + ((u64)q * 5 + k) * m_in + 2 * i);
                        a[q].v[k] = v.x;
                        b[q].v[k] = v.y;
                    }
            } else {
                // storage: n_l(x) = n[2x], n_r(x) = n[2x+1]; inputs x = 2i, 2i+1 -> entries 4i .. 4i+3
                const u64 plane = 2 * m_in;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    uint4 v = *reinterpret_cast<const uint4*>(d_in + (u64)k * plane + 4 * i);
                    a[2].v[k] = v.x;
                    a[3].v[k] = v.y;
                    b[2].v[k] = v.z;
                    b[3].v[k] = v.w;
                }
                if (MODE == 0) {
                    uint4 v = *reinterpret_cast<const uint4*>(n_in + 4 * i);
                    a[0] = ef_from_base(v.x);
                    a[1] = ef_from_base(v.y);
                    b[0] = ef_from_base(v.z);
                    b[1] = ef_from_base(v.w);
                } else {
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        uint4 v = *reinterpret_cast<const uint4*>(n_in + (u64)k * plane + 4 * i);
                        a[0].v[k] = v.x;
                        a[1].v[k] = v.y;
                        b[0].v[k] = v.z;
                        b[1].v[k] = v.w;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (MODE == 0 && q < 2) {
                    // base numerators: r * (b - a) is EF x base
                    EF t2 = ef_mul_base(r, sub(b[q].v[0], a[q].v[0]));
                    t2.v[0] = add(t2.v[0], a[q].v[0]);
                    o[t][q] = t2;
                } else {
                    o[t][q] = ef_add(a[q], ef_mul(r, ef_sub(b[q], a[q])));
                }
#pragma unroll
                for (int k = 0; k < 5; k++) out[((u64)q * 5 + k) * m_out + i] = o[t][q].v[k];
            }
        }
        if constexpr (!LAST) {
            const EF w = eq_split_at(eq, j);
            pair_accumulate(o[0][0], o[N_OUT - 1][0], o[0][1], o[N_OUT - 1][1], o[0][2], o[N_OUT - 1][2], o[0][3], o[N_OUT - 1][3], w, acc);
        }
    }
    if constexpr (!LAST) {
        finish_round(acc, lds, tot, partial, alpha, final_out, seq);
    } else {
        // layer end: the four folded values go straight to the pinned result buffer (one lane wrote them)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int k = 0; k < 5; k++) final_out[q * 5 + k] = out[((u64)q * 5 + k) * m_out];
            lm_publish_flag(final_out, seq);
        }
    }
}

extern "C" {

void lm_gkr_free(lm_ctx* ctx, lm_gkr* g) {
    if (!g) return;
    for (u32* p : g->nums) lm_pool_free(ctx, p);
    for (u32* p : g->dens) lm_pool_free(ctx, p);
    for (int i = 0; i < 2; i++) lm_pool_free(ctx, g->work[i]);
    lm_pool_free(ctx, g->eqt.d_buf);
    delete g;
}

int lm_gkr_build(lm_ctx* ctx, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars, lm_gkr** out) {
    LM_REQUIRE(ctx && d_nums && d_dens && out && n_vars > 5 && n_vars <= 30);
    lm_gkr* g = new lm_gkr();
    g->n_vars = n_vars;
    g->d_nums0 = d_nums;
    g->d_dens0 = d_dens;
    const u32* n_in = d_nums;
    const u32* d_in = d_dens;
    for (u32 v = n_vars - 1; v >= 5; v--) {
        const u64 m = 1ull << v;
        u32 *nn = nullptr, *dd = nullptr;
        if (lm_pool_alloc_t(ctx, &nn, 5 * m * 4) != hipSuccess || lm_pool_alloc_t(ctx, &dd, 5 * m * 4) != hipSuccess) {
            lm_set_error("lm_gkr_build: device allocation failed");
            lm_pool_free(ctx, nn);
            lm_gkr_free(ctx, g);
            return LM_E_NOMEM;
        }
        g->nums.push_back(nn);
        g->dens.push_back(dd);
        const u32 blocks = (u32)std::min<u64>((m + 255) / 256, 4096);
        if (v == n_vars - 1)
            LM_LAUNCH(ctx, k_gkr_layer_up<true>, dim3(blocks), dim3(256), 0, n_in, d_in, m, nn, dd);
        else
            LM_LAUNCH(ctx, k_gkr_layer_up<false>, dim3(blocks), dim3(256), 0, n_in, d_in, m, nn, dd);
        n_in = nn;
        d_in = dd;
    }
    // work buffers: first fold of the biggest layer yields 4 arrays of 2^(n_vars-2) EF
    g->work_words = 20ull << (n_vars - 2);
    const u64 w1 = std::max<u64>(g->work_words / 2, 64);
    if (lm_pool_alloc_t(ctx, &g->work[0], g->work_words * 4) != hipSuccess ||
        lm_pool_alloc_t(ctx, &g->work[1], w1 * 4) != hipSuccess ||
        lm_pool_alloc_t(ctx, &g->eqt.d_buf, PrefixEqTables::words_needed(n_vars) * 4) != hipSuccess) {
        lm_set_error("lm_gkr_build: device allocation failed (work)");
        lm_gkr_free(ctx, g);
        return LM_E_NOMEM;
    }
    g->eqt.buf_words = PrefixEqTables::words_needed(n_vars);
    LM_HIP(hipGetLastError());
    *out = g;
    return LM_OK;
}

int lm_gkr_top(lm_ctx* ctx, const lm_gkr* g, uint32_t* nums32, uint32_t* dens32) {
    LM_REQUIRE(ctx && g && nums32 && dens32);
    u32 soa[2][160];
    LM_HIP(hipMemcpyAsync(soa[0], g->nums.back(), 640, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(hipMemcpyAsync(soa[1], g->dens.back(), 640, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 32; i++)
        for (int k = 0; k < 5; k++) {
            nums32[i * 5 + k] = soa[0][k * 32 + i];
            dens32[i * 5 + k] = soa[1][k * 32 + i];
        }
    return LM_OK;
}

// Start the sumcheck of the layer with 2^(K+1) entries (K = number of coordinates of the claim point, 5 <= K < n_vars).
int lm_gkr_layer_begin(lm_ctx* ctx, lm_gkr* g, uint32_t K, const uint32_t* point, const uint32_t alpha[5]) {
    LM_REQUIRE(ctx && g && point && alpha && K >= 5 && K < g->n_vars);
    g->K = K;
    g->round = 0;
    g->cur = -1;
    g->m = 1ull << K;  // length of each of n_l, n_r, d_l, d_r
    memcpy(g->alpha.v, alpha, 20);
    // round t uses eq over point[0 .. p), p = K-1-t
    return g->eqt.build(ctx, point, K);
}

// One round.  prev_r = NULL on the first round of the layer; afterwards the challenge of the previous round (the
// arrays are folded by it first).  out = (c0_raw, c2_raw) of finalize_round (sumcheck_utils.rs:90-109), padding included
// because the vectors are fully materialised.
int lm_gkr_round(lm_ctx* ctx, lm_gkr* g, const uint32_t* prev_r, uint32_t out_c0_c2[10]) {
    LM_REQUIRE(ctx && g && out_c0_c2 && g->round < g->K);
    LM_REQUIRE((g->round == 0) == (prev_r == nullptr));
    const u32 t = g->round;
    const u32 p = g->K - 1 - t;  // prefix coordinates of this round's eq table
    const u64 n_pairs = 1ull << p;
    // up to 4096 pairs: one workgroup does the whole round (no partials, no second launch)
    const u32 blocks = n_pairs <= 512 ? 1 : (u32)std::min<u64>((n_pairs + 255) / 256, 1024);
    const u32 seq = ++ctx->res_seq;
    u32* s;
    int rc = lm_scratch(ctx, (u64)blocks * 20 + 32, &s);
    if (rc) return rc;
    u32* d_out = s + (u64)blocks * 20;
    const EqSplit eq = g->eqt.at(p);
    // the layer being proven has 2^(K+1) entries: the caller's input when K + 1 == n_vars, else owned layer
    // nums[i] (2^(n_vars-1-i) entries) with i = n_vars - K - 2
    const bool input_layer = g->K == g->n_vars - 1;
    const u32* n_st = input_layer ? g->d_nums0 : g->nums[g->n_vars - g->K - 2];
    const u32* d_st = input_layer ? g->d_dens0 : g->dens[g->n_vars - g->K - 2];
    if (t == 0) {
        if (input_layer)
            LM_LAUNCH(ctx, k_gkr_round_storage<true>, dim3(blocks), dim3(256), 0, n_st, d_st, n_pairs, eq, s, g->alpha, ctx->h_res, seq);
        else
            LM_LAUNCH(ctx, k_gkr_round_storage<false>, dim3(blocks), dim3(256), 0, n_st, d_st, n_pairs, eq, s, g->alpha, ctx->h_res, seq);
    } else {
        EF r;
        memcpy(r.v, prev_r, 20);
        const u64 m_out = g->m / 2;
        const int dst = g->cur < 0 ? 0 : 1 - g->cur;
        if (g->cur < 0) {
            if (input_layer)
                LM_LAUNCH(ctx, (k_gkr_fold_round<0, false>), dim3(blocks), dim3(256), 0, n_st, d_st, (const u32*)nullptr, m_out, r, eq,
                          g->work[dst], s, g->alpha, ctx->h_res, seq);
            else
                LM_LAUNCH(ctx, (k_gkr_fold_round<1, false>), dim3(blocks), dim3(256), 0, n_st, d_st, (const u32*)nullptr, m_out, r, eq,
                          g->work[dst], s, g->alpha, ctx->h_res, seq);
        } else {
            LM_LAUNCH(ctx, (k_gkr_fold_round<2, false>), dim3(blocks), dim3(256), 0, (const u32*)nullptr, (const u32*)nullptr,
                      (const u32*)g->work[g->cur], m_out, r, eq, g->work[dst], s, g->alpha, ctx->h_res, seq);
        }
        g->cur = dst;
        g->m = m_out;
    }
    (void)d_out;
    if (blocks > 1) LM_LAUNCH(ctx, k_gkr_reduce, dim3(1), dim3(256), 0, (const u32*)s, blocks, g->alpha, ctx->h_res, seq);
    LM_HIP(hipGetLastError());
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(out_c0_c2, ctx->h_res, 40);
    g->round++;
    return LM_OK;
}

// After the last round: fold by the last challenge and return [n_l, n_r, d_l, d_r] (4 EF, mod.rs:129).
int lm_gkr_layer_end(lm_ctx* ctx, lm_gkr* g, const uint32_t last_r[5], uint32_t inner_evals[20]) {
    LM_REQUIRE(ctx && g && last_r && inner_evals && g->round == g->K && g->m == 2);
    EF r;
    memcpy(r.v, last_r, 20);
    const int dst = 1 - g->cur;
    EqSplit eq = g->eqt.at(0);
    u32* s;
    int rc = lm_scratch(ctx, 64, &s);
    if (rc) return rc;
    LM_REQUIRE(g->cur >= 0);  // K >= 5 rounds, so at least one fold happened
    const u32 seq = ++ctx->res_seq;
    LM_LAUNCH(ctx, (k_gkr_fold_round<2, true>), dim3(1), dim3(256), 0, (const u32*)nullptr, (const u32*)nullptr,
              (const u32*)g->work[g->cur], (u64)1, r, eq, g->work[dst], s, g->alpha, ctx->h_res, seq);
    LM_HIP(hipGetLastError());
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(inner_evals, ctx->h_res, 80);
    g->cur = dst;
    g->m = 1;
    return LM_OK;
}

}  // extern "C"
