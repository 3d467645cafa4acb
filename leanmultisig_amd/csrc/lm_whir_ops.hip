// Device side of WhirConfig::prove: weight-polynomial construction, product-sumcheck rounds, folds, PoW grinding.
#include <stdlib.h>
#include <algorithm>
#include "lm_common.h"
#include "poseidon16_quad.h"

using namespace kb;

// =====================================================================================================
// Weights.  W[offset + i] += sum_{items of the group} scalar_j * w_j(i)
//   eq   : w(i) = eq(point, i) = T_hi[i >> k_lo] * T_lo[i & mask]   (scalar folded into T_hi)
//   next : w(i) = eq(point, i - 1) for i >= 1, plus eq(point, 2^n - 1) at i = 2^n - 1
//          (matrix_next_mle_folded, crates/backend/poly/src/next_mle.rs:35-53: weight of x = i - 1, wrap-around at the
//          all-ones index)
// =====================================================================================================
static constexpr u32 W_KLO = 10;
static constexpr u32 W_CHUNK = 1u << W_KLO;
static constexpr u32 W_FAST_TILE = 256;  // fast items staged in LDS per pass

struct WItem {
    u64 thi_off, tlo_off;  // word offsets into the table arena (SoA tables)
    u32 inner_n, is_next;
    u64 point_off;         // EF index into the points array
    u32 lo_base, pad;      // 1: the last k_lo coordinates are base-field elements, T_lo has only plane 0 (STIR query points)
};
struct WGroup {
    u64 offset;
    u64 chunk_begin;
    u32 inner_n, item_begin, item_end;
    u32 fast_end;  // items [item_begin, fast_end) take the base-point path
};

// grid: (blk_pre[n_items], 2): item k owns the workgroups [blk_pre[k], blk_pre[k + 1]) — as many as its longer table needs.  (As a
// (ceil(max_table / 256), n_items) grid, the ~270 items of the opening's statement — two of them with 2^16-entry tables — were 138 k
// workgroups of which 99 % left at once: 92 us of dispatch.)  y = 0: T_hi over the first inner - k_lo coordinates, times the scalar;
// y = 1: T_lo over the last k_lo coordinates.
__global__ __launch_bounds__(256) void k_weight_tables(const WItem* __restrict__ items, const u32* __restrict__ points,
                                                       const u32* __restrict__ scalars, u32* __restrict__ arena,
                                                       const u32* __restrict__ blk_pre, u32 n_items) {
    u32 a = 0, b = n_items;  // the item whose range holds blockIdx.x (uniform: scalar loads, <= 12 steps)
    while (b - a > 1) {
        const u32 m = (a + b) >> 1;
        if (blk_pre[m] <= blockIdx.x)
            a = m;
        else
            b = m;
    }
    const u32 item = a;
    const WItem it = items[item];
    const u32 k_lo = it.inner_n < W_KLO ? it.inner_n : W_KLO;
    const u32 k_hi = it.inner_n - k_lo;
    const bool lo = blockIdx.y == 1;
    const u32 nb = lo ? k_lo : k_hi;
    const u32 len = 1u << nb;
    const u32 i = (blockIdx.x - blk_pre[item]) * 256 + threadIdx.x;
    if (i >= len) return;
    const u32* pt = points + (it.point_off + (lo ? k_hi : 0)) * 5;
    EF acc;
    if (lo) {
        acc = ef_one();
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) acc.v[k] = scalars[item * 5 + k];
    }
    for (u32 j = 0; j < nb; j++) {
        EF p;
#pragma unroll
        for (int k = 0; k < 5; k++) p.v[k] = pt[j * 5 + k];
        u32 bit = (i >> (nb - 1 - j)) & 1;
        acc = ef_mul(acc, bit ? p : ef_sub(ef_one(), p));
    }
    u32* dst = arena + (lo ? it.tlo_off : it.thi_off);
#pragma unroll
    for (int k = 0; k < 5; k++) dst[(u64)k * len + i] = acc.v[k];
}

__device__ __forceinline__ EF weight_eq_at(const u32* __restrict__ arena, const WItem& it, u32 k_lo, u64 j) {
    const u64 hi_len = 1ull << (it.inner_n - k_lo);
    const u32 lo_len = 1u << k_lo;
    const u64 jh = j >> k_lo;
    const u32 jl = (u32)j & (lo_len - 1);
    EF a, b;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        a.v[k] = arena[it.thi_off + (u64)k * hi_len + jh];
        b.v[k] = arena[it.tlo_off + (u64)k * lo_len + jl];
    }
    return ef_mul(a, b);
}

// Contribution of one group to the U elements of this lane: element u is group-relative index i_base + u*256 + tid.
// acc[u] += sum_items scalar * w(i).  Block-uniform control flow (contains __syncthreads).
static constexpr u32 W_U = W_CHUNK / 256;  // elements per lane
__device__ __forceinline__ void group_contrib(const WGroup& g, u64 i_base, const WItem* __restrict__ items,
                                              const u32* __restrict__ arena, u32* a_lds, EF acc[W_U]) {
    const u64 len = 1ull << g.inner_n;
    const u32 k_lo = g.inner_n < W_KLO ? g.inner_n : W_KLO;
    const u64 hi_len = 1ull << (g.inner_n - k_lo);
    const u32 lo_len = 1u << k_lo;
    const u64 jh = i_base >> k_lo;  // uniform: a chunk never straddles a T_hi entry (W_CHUNK = 2^W_KLO)
    // ---- fast items [item_begin, fast_end): eq weights whose T_lo is a base-field table (STIR query points; the later
    // rounds add hundreds of them to one region).  The group's T_hi entries for this chunk are staged in LDS once, then
    // each item costs one LDS broadcast, U coalesced loads and 5 U multiply-adds into 64-bit accumulators.
    const u32 n_fast = g.fast_end - g.item_begin;
    if (n_fast) {
        u64 a64[W_U][5];
#pragma unroll
        for (u32 u = 0; u < W_U; u++)
#pragma unroll
            for (int k = 0; k < 5; k++) a64[u][k] = 0;
        const WItem first = items[g.item_begin];
        const u64 stride = 5 * hi_len + 5 * (u64)lo_len;  // arena layout of consecutive items of one group
        bool live[W_U];
        u32 jl[W_U];
#pragma unroll
        for (u32 u = 0; u < W_U; u++) {
            const u64 i = i_base + u * 256 + threadIdx.x;
            live[u] = i < len;
            jl[u] = (u32)i & (lo_len - 1);
        }
        u32 pending = 0;
        for (u32 t0 = 0; t0 < n_fast; t0 += W_FAST_TILE) {
            const u32 nt = n_fast - t0 < W_FAST_TILE ? n_fast - t0 : W_FAST_TILE;
            __syncthreads();
            for (u32 x = threadIdx.x; x < nt * 5; x += 256) {
                const u32 t = x / 5, k = x - t * 5;
                a_lds[x] = arena[first.thi_off + (u64)(t0 + t) * stride + (u64)k * hi_len + jh];
            }
            __syncthreads();
            const u32* lo_tab = arena + first.tlo_off + (u64)t0 * stride;
#pragma unroll 2
            for (u32 t = 0; t < nt; t++) {
                u32 bl[W_U];
#pragma unroll
                for (u32 u = 0; u < W_U; u++) bl[u] = live[u] ? lo_tab[(u64)t * stride + jl[u]] : 0u;
                u32 av[5];
#pragma unroll
                for (int k = 0; k < 5; k++) av[k] = a_lds[t * 5 + k];
#pragma unroll
                for (u32 u = 0; u < W_U; u++)
#pragma unroll
                    for (int k = 0; k < 5; k++) a64[u][k] += (u64)av[k] * bl[u];
                if (++pending == 3) {
                    pending = 0;
#pragma unroll
                    for (u32 u = 0; u < W_U; u++)
#pragma unroll
                        for (int k = 0; k < 5; k++) a64[u][k] = fold32(a64[u][k]);
                }
            }
        }
#pragma unroll
        for (u32 u = 0; u < W_U; u++)
#pragma unroll
            for (int k = 0; k < 5; k++) acc[u].v[k] = add(acc[u].v[k], reduce(fold32(a64[u][k])));
    }
    // ---- remaining items (EF points, `next` weights)
    if (g.fast_end == g.item_end) return;
#pragma unroll
    for (u32 u = 0; u < W_U; u++) {
        const u64 i = i_base + u * 256 + threadIdx.x;
        if (i >= len) continue;
        for (u32 j = g.fast_end; j < g.item_end; j++) {
            const WItem it = items[j];
            if (!it.is_next) {
                acc[u] = ef_add(acc[u], weight_eq_at(arena, it, k_lo, i));
            } else {
                if (i >= 1) acc[u] = ef_add(acc[u], weight_eq_at(arena, it, k_lo, i - 1));
                if (i == len - 1) acc[u] = ef_add(acc[u], weight_eq_at(arena, it, k_lo, i));
            }
        }
    }
}

// W[g.offset + i] += contributions of g, one launch for all groups of one inner_n (disjoint regions).
__global__ __launch_bounds__(256) void k_weights_accumulate(u32* __restrict__ W, u64 plane, const WGroup* __restrict__ groups,
                                                            u32 n_groups, const WItem* __restrict__ items,
                                                            const u32* __restrict__ arena) {
    __shared__ u32 a_lds[W_FAST_TILE * 5];
    // find the group of this block
    u32 lo = 0, hi = n_groups - 1;
    const u64 b = blockIdx.x;
    while (lo < hi) {
        u32 mid = (lo + hi + 1) >> 1;
        if (groups[mid].chunk_begin <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const WGroup g = groups[lo];
    const u64 i_base = (b - g.chunk_begin) * W_CHUNK;
    EF acc[W_U];
#pragma unroll
    for (u32 u = 0; u < W_U; u++) acc[u] = ef_zero();
    group_contrib(g, i_base, items, arena, a_lds, acc);
#pragma unroll
    for (u32 u = 0; u < W_U; u++) {
        const u64 i = i_base + u * 256 + threadIdx.x;
        if (i >= (1ull << g.inner_n)) continue;
        u32* w = W + g.offset + i;
#pragma unroll
        for (int k = 0; k < 5; k++) w[(u64)k * plane] = add(w[(u64)k * plane], acc[u].v[k]);
    }
}

// W <- (whole-domain group) + (the one top-level region that contains the chunk, if any): plain stores, W is written once.
// groups[0] = the whole-domain group (possibly without items), groups[1..n_groups) = disjoint regions of >= W_CHUNK
// elements sorted by offset.  One block per chunk of the domain.
__global__ __launch_bounds__(256) void k_weights_init(u32* __restrict__ W, u64 plane, const WGroup* __restrict__ groups,
                                                      u32 n_groups, const WItem* __restrict__ items,
                                                      const u32* __restrict__ arena) {
    __shared__ u32 a_lds[W_FAST_TILE * 5];
    const u64 i_base = (u64)blockIdx.x * W_CHUNK;
    EF acc[W_U];
#pragma unroll
    for (u32 u = 0; u < W_U; u++) acc[u] = ef_zero();
    group_contrib(groups[0], i_base, items, arena, a_lds, acc);
    if (n_groups > 1) {  // last region with offset <= i_base
        u32 lo = 1, hi = n_groups - 1;
        while (lo < hi) {
            u32 mid = (lo + hi + 1) >> 1;
            if (groups[mid].offset <= i_base)
                lo = mid;
            else
                hi = mid - 1;
        }
        const WGroup g = groups[lo];
        if (g.offset <= i_base && i_base < g.offset + (1ull << g.inner_n)) group_contrib(g, i_base - g.offset, items, arena, a_lds, acc);
    }
#pragma unroll
    for (u32 u = 0; u < W_U; u++) {
        const u64 i = i_base + u * 256 + threadIdx.x;
        if (i >= plane) continue;
#pragma unroll
        for (int k = 0; k < 5; k++) W[(u64)k * plane + i] = acc[u].v[k];
    }
}

// =====================================================================================================
// Product sumcheck round: partial sums per block, then one reducing block.
// =====================================================================================================
__device__ __forceinline__ u64 fold64(u64 t) {
    u64 y = t - P_SHL32;
    return t >= P_SHL32 ? y : t;
}

__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = add(v, (u32)__shfl_down(v, off, 64));
    return v;
}
// sums 10 field values across the block; thread 0 writes them to dst[0..10)
__device__ __forceinline__ void block_sum10(u32 v[10], u32* lds /* 40 words */, u32* dst, u32* publish_to = nullptr, u32 seq = 0) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; k++) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) lds[wave * 10 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        u32 s = 0;
        for (u32 w = 0; w < (blockDim.x >> 6); w++) s = add(s, lds[w * 10 + threadIdx.x]);
        if (publish_to)
            lm_store_system(dst + threadIdx.x, s);
        else
            dst[threadIdx.x] = s;
    }
    if (publish_to) {  // all ten writers are in wave 0, which waits for its stores before one lane stores the sequence number
        if (threadIdx.x < 64) {
            lm_wait_stores();
            if (threadIdx.x == 0) lm_publish_flag(publish_to, seq);
        }
    }
}

// End of a product-sumcheck round kernel: a single block publishes its sums; otherwise every block adds its 10 words into
// the context's accumulators and the block that finishes last publishes the totals (lm_grid_sum: no reducing launch).
__device__ __forceinline__ void finish10(u32 v[10], u32* red /* 64 words */, unsigned long long* __restrict__ acc,
                                         u32* __restrict__ done_counter, u32* __restrict__ h_res, u32 seq) {
    if (gridDim.x == 1) {
        block_sum10(v, red, h_res, h_res, seq);
        return;
    }
    block_sum10(v, red, red + 40);
    __syncthreads();
    if (!lm_grid_sum<10>(red + 40, acc, done_counter, red)) return;
    if (threadIdx.x < 64) {
        if (threadIdx.x < 10) lm_store_system(h_res + threadIdx.x, red[threadIdx.x]);
        lm_wait_stores();
        if (threadIdx.x == 0) lm_publish_flag(h_res, seq);
    }
}

__global__ __launch_bounds__(256) void k_prod_round_base(const u32* __restrict__ f, const u32* __restrict__ W, u64 half,
                                                         unsigned long long* __restrict__ acc, u32* __restrict__ done_counter,
                                                         u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 red[64];
    const u64 plane = 2 * half;
    u64 a0[5] = {0, 0, 0, 0, 0}, a2[5] = {0, 0, 0, 0, 0};
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < half; i += (u64)gridDim.x * 256) {
        const u32 f0 = f[i], f1 = f[i + half];
        const u32 df = sub(f1, f0);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const u32 w0 = W[(u64)k * plane + i], w1 = W[(u64)k * plane + i + half];
            a0[k] = fold64(a0[k] + (u64)f0 * w0);
            a2[k] = fold64(a2[k] + (u64)df * sub(w1, w0));
        }
    }
    u32 v[10];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        v[k] = reduce(a0[k]);
        v[5 + k] = reduce(a2[k]);
    }
    finish10(v, red, acc, done_counter, final_out, seq);
}
__global__ __launch_bounds__(256) void k_prod_round_ext(const u32* __restrict__ f, const u32* __restrict__ W, u64 half,
                                                        unsigned long long* __restrict__ acc, u32* __restrict__ done_counter,
                                                        u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 red[64];
    const u64 plane = 2 * half;
    EF c0 = ef_zero(), c2 = ef_zero();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < half; i += (u64)gridDim.x * 256) {
        EF f0, f1, w0, w1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            f0.v[k] = f[(u64)k * plane + i];
            f1.v[k] = f[(u64)k * plane + i + half];
            w0.v[k] = W[(u64)k * plane + i];
            w1.v[k] = W[(u64)k * plane + i + half];
        }
        c0 = ef_add(c0, ef_mul(f0, w0));
        c2 = ef_add(c2, ef_mul(ef_sub(f1, f0), ef_sub(w1, w0)));
    }
    u32 v[10];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        v[k] = c0.v[k];
        v[5 + k] = c2.v[k];
    }
    finish10(v, red, acc, done_counter, final_out, seq);
}
__global__ __launch_bounds__(256) void k_fold_base(const u32* __restrict__ in, u64 half, EF r, u32* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < half; i += (u64)gridDim.x * 256) {
        const u32 a = in[i], d = sub(in[i + half], a);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            u32 t = mul(r.v[k], d);
            out[(u64)k * half + i] = k == 0 ? add(t, a) : t;
        }
    }
}
__global__ __launch_bounds__(256) void k_fold_ext(const u32* __restrict__ in, u64 half, EF r, u32* __restrict__ out) {
    const u64 plane = 2 * half;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < half; i += (u64)gridDim.x * 256) {
        EF a, b;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            a.v[k] = in[(u64)k * plane + i];
            b.v[k] = in[(u64)k * plane + i + half];
        }
        EF o = ef_add(a, ef_mul(r, ef_sub(b, a)));
#pragma unroll
        for (int k = 0; k < 5; k++) out[(u64)k * half + i] = o.v[k];
    }
}

// Fold both tables by r and accumulate the NEXT round's (c0, c2) on the folded values while they are in registers:
// one pass instead of fold(f), fold(W), prod_round (which would read f', W' again).  quarter = half / 2 >= 1.
// Lane i < quarter produces outputs i and i + quarter (the next round's pair) from inputs i, i+quarter, i+half, i+half+quarter.
template <bool F_BASE>
__global__ __launch_bounds__(256) void k_fold_round(const u32* __restrict__ f, const u32* __restrict__ W, u64 half, EF r,
                                                    u32* __restrict__ f_out, u32* __restrict__ W_out, unsigned long long* __restrict__ acc,
                                                    u32* __restrict__ done_counter, u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 red[64];
    const u64 plane = 2 * half, quarter = half >> 1;
    EF c0 = ef_zero(), c2 = ef_zero();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < quarter; i += (u64)gridDim.x * 256) {
        EF fo[2], wo[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const u64 j = i + t * quarter;
            EF a, b;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                a.v[k] = W[(u64)k * plane + j];
                b.v[k] = W[(u64)k * plane + j + half];
            }
            wo[t] = ef_add(a, ef_mul(r, ef_sub(b, a)));
            if (F_BASE) {
                const u32 fa = f[j], d = sub(f[j + half], fa);
                fo[t] = ef_mul_base(r, d);
                fo[t].v[0] = add(fo[t].v[0], fa);
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    a.v[k] = f[(u64)k * plane + j];
                    b.v[k] = f[(u64)k * plane + j + half];
                }
                fo[t] = ef_add(a, ef_mul(r, ef_sub(b, a)));
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                f_out[(u64)k * half + j] = fo[t].v[k];
                W_out[(u64)k * half + j] = wo[t].v[k];
            }
        }
        c0 = ef_add(c0, ef_mul(fo[0], wo[0]));
        c2 = ef_add(c2, ef_mul(ef_sub(fo[1], fo[0]), ef_sub(wo[1], wo[0])));
    }
    u32 v[10];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        v[k] = c0.v[k];
        v[5 + k] = c2.v[k];
    }
    finish10(v, red, acc, done_counter, final_out, seq);
}

// =====================================================================================================
// Two product-sumcheck rounds per pass (MSB-first folding: round t pairs i with i + n/2, round t+1 pairs i with i + n/4).
// Over a quad (x00, x01, x10, x11) = x[i], x[i + n/4], x[i + n/2], x[i + 3n/4], i < n/4, with D0 = x10 - x00, D1 = x11 - x01:
//   round t   : c0 = sum f00 W00 + f01 W01                c2 = sum D0f D0W + D1f D1W
//   round t+1 , after folding by the challenge r of round t (x'0 = x00 + r D0, x'1 = x01 + r D1):
//               c0'(r) = P00 + r (P10 - P00 - Q0) + r^2 Q0         P00 = sum f00 W00, P10 = sum f10 W10, Q0 = sum D0f D0W
//               c2'(r) = T0 + r (T3 - T0 - T2) + r^2 T2            T0 / T3 / T2 = sum of (f, W) products of x01 - x00 / x11 - x10 / D1 - D0
// Eight sums (P00, P01, P10, Q0, Q1, T0, T2, T3; 40 words) give both rounds: the host evaluates the quadratics at r
// (lm_host.cpp: sumcheck_rounds) and the next pass folds by two challenges at once — half the passes over f and W, half
// the host round trips.  Exact field identities: the transcript is unchanged.
// =====================================================================================================
static constexpr int PR2_WORDS = 40;
struct Quad2 {
    EF p00, p01, p10, q0, q1, t0, t2, t3;
};
__device__ __forceinline__ void quad_accumulate(const EF f[4], const EF w[4], Quad2& a) {  // order: 00, 01, 10, 11
    const EF d0f = ef_sub(f[2], f[0]), d1f = ef_sub(f[3], f[1]), d0w = ef_sub(w[2], w[0]), d1w = ef_sub(w[3], w[1]);
    a.p00 = ef_add(a.p00, ef_mul(f[0], w[0]));
    a.p01 = ef_add(a.p01, ef_mul(f[1], w[1]));
    a.p10 = ef_add(a.p10, ef_mul(f[2], w[2]));
    a.q0 = ef_add(a.q0, ef_mul(d0f, d0w));
    a.q1 = ef_add(a.q1, ef_mul(d1f, d1w));
    a.t0 = ef_add(a.t0, ef_mul(ef_sub(f[1], f[0]), ef_sub(w[1], w[0])));
    a.t3 = ef_add(a.t3, ef_mul(ef_sub(f[3], f[2]), ef_sub(w[3], w[2])));
    a.t2 = ef_add(a.t2, ef_mul(ef_sub(d1f, d0f), ef_sub(d1w, d0w)));
}
__device__ __forceinline__ void quad_accumulate_base(const u32 f[4], const EF w[4], Quad2& a) {
    const u32 d0f = sub(f[2], f[0]), d1f = sub(f[3], f[1]);
    const EF d0w = ef_sub(w[2], w[0]), d1w = ef_sub(w[3], w[1]);
    a.p00 = ef_add(a.p00, ef_mul_base(w[0], f[0]));
    a.p01 = ef_add(a.p01, ef_mul_base(w[1], f[1]));
    a.p10 = ef_add(a.p10, ef_mul_base(w[2], f[2]));
    a.q0 = ef_add(a.q0, ef_mul_base(d0w, d0f));
    a.q1 = ef_add(a.q1, ef_mul_base(d1w, d1f));
    a.t0 = ef_add(a.t0, ef_mul_base(ef_sub(w[1], w[0]), sub(f[1], f[0])));
    a.t3 = ef_add(a.t3, ef_mul_base(ef_sub(w[3], w[2]), sub(f[3], f[2])));
    a.t2 = ef_add(a.t2, ef_mul_base(ef_sub(d1w, d0w), sub(d1f, d0f)));
}
// block sum of the 40 words, then the grid sum / publication (like finish10)
__device__ __forceinline__ void finish40(const Quad2& a, u32* red /* 4 * 40 + 40 words */, unsigned long long* __restrict__ acc,
                                         u32* __restrict__ done_counter, u32* __restrict__ h_res, u32 seq) {
    const EF* e = &a.p00;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const u32 v = wave_sum(e[q].v[k]);
            if (lane == 0) red[wave * PR2_WORDS + q * 5 + k] = v;
        }
    __syncthreads();
    u32* tot = red + 4 * PR2_WORDS;
    if (threadIdx.x < PR2_WORDS) {
        u32 t = 0;
        for (u32 w = 0; w < (blockDim.x >> 6); w++) t = add(t, red[w * PR2_WORDS + threadIdx.x]);
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (gridDim.x > 1 && !lm_grid_sum<PR2_WORDS>(tot, acc, done_counter, tot)) return;
    if (threadIdx.x < 64) {
        if (threadIdx.x < PR2_WORDS) lm_store_system(h_res + threadIdx.x, tot[threadIdx.x]);
        lm_wait_stores();
        if (threadIdx.x == 0) lm_publish_flag(h_res, seq);
    }
}
// Base-field f (the first pass of the opening sumcheck, 1.6 GB): the eight sums as 40 raw 64-bit accumulators — a product of
// reduced values is < p^2 < 2^62, kb::fold32 brings an accumulator below 2^57 with one multiply-add, after which three more
// products fit — reduced once at the end.  ~180 VALU instructions per quad instead of ~600: the pass is as much an ALU kernel as
// a memory kernel (6 instructions per byte with a Montgomery reduction per product).
struct Quad2Raw {
    u64 s[8][5];  // order of Quad2: p00, p01, p10, q0, q1, t0, t2, t3
};
__device__ __forceinline__ void quad_accumulate_base_raw(const u32 f[4], const EF w[4], Quad2Raw& a) {
    const u32 d0f = sub(f[2], f[0]), d1f = sub(f[3], f[1]);
    const EF d0w = ef_sub(w[2], w[0]), d1w = ef_sub(w[3], w[1]);
    const EF t0w = ef_sub(w[1], w[0]), t3w = ef_sub(w[3], w[2]), t2w = ef_sub(d1w, d0w);
    const u32 t0f = sub(f[1], f[0]), t3f = sub(f[3], f[2]), t2f = sub(d1f, d0f);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        a.s[0][k] += (u64)w[0].v[k] * f[0];
        a.s[1][k] += (u64)w[1].v[k] * f[1];
        a.s[2][k] += (u64)w[2].v[k] * f[2];
        a.s[3][k] += (u64)d0w.v[k] * d0f;
        a.s[4][k] += (u64)d1w.v[k] * d1f;
        a.s[5][k] += (u64)t0w.v[k] * t0f;
        a.s[6][k] += (u64)t2w.v[k] * t2f;
        a.s[7][k] += (u64)t3w.v[k] * t3f;
    }
}
// the eight sums on the tables as they are (no fold): lane i < n/4
template <bool F_BASE>
__global__ __launch_bounds__(256) void k_prod_round2(const u32* __restrict__ f, const u32* __restrict__ W, u64 quarter,
                                                     unsigned long long* __restrict__ acc, u32* __restrict__ done_counter,
                                                     u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 red[5 * PR2_WORDS];
    const u64 plane = 4 * quarter;
    Quad2 a;
    a.p00 = a.p01 = a.p10 = a.q0 = a.q1 = a.t0 = a.t2 = a.t3 = ef_zero();
    Quad2Raw raw;
    if (F_BASE) {
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
            for (int k = 0; k < 5; k++) raw.s[q][k] = 0;
    }
    u32 room = 4;  // products that still fit the raw accumulators
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < quarter; i += (u64)gridDim.x * 256) {
        EF w[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < 5; k++) w[t].v[k] = W[(u64)k * plane + i + t * quarter];
        if (F_BASE) {
            u32 fb[4];
#pragma unroll
            for (int t = 0; t < 4; t++) fb[t] = f[i + t * quarter];
            if (room == 0) {
#pragma unroll
                for (int q = 0; q < 8; q++)
#pragma unroll
                    for (int k = 0; k < 5; k++) raw.s[q][k] = fold32(raw.s[q][k]);
                room = 3;
            }
            quad_accumulate_base_raw(fb, w, raw);
            room--;
        } else {
            EF fe[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int k = 0; k < 5; k++) fe[t].v[k] = f[(u64)k * plane + i + t * quarter];
            quad_accumulate(fe, w, a);
        }
    }
    if (F_BASE) {
        EF* e = &a.p00;
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
            for (int k = 0; k < 5; k++) e[q].v[k] = reduce(fold32(raw.s[q][k]));
    }
    finish40(a, red, acc, done_counter, final_out, seq);
}
// Fold f and W by two challenges (r0: the variable pairing i with i + n/2, then r1: i with i + n/4) and, on the folded
// tables of m = n/4 entries, compute the eight sums (SUMS = 2, needs m >= 4), only (c0, c2) of the next round (SUMS = 1,
// m >= 2: words 0..9 of the result) or nothing (SUMS = 0).  Lane j < max(m/4, 1) produces outputs j + t * (m/4), t < 4.
template <bool F_BASE, int SUMS>
__global__ __launch_bounds__(256) void k_fold2_round(const u32* __restrict__ f, const u32* __restrict__ W, u64 m, EF r0, EF r1,
                                                     u32* __restrict__ f_out, u32* __restrict__ W_out,
                                                     unsigned long long* __restrict__ acc, u32* __restrict__ done_counter,
                                                     u32* __restrict__ final_out, u32 seq) {
    __shared__ u32 red[5 * PR2_WORDS];
    const u64 plane = 4 * m;                      // input entries per plane; inputs of output x: x, x + m, x + 2m, x + 3m
    const u64 step = m >= 4 ? m / 4 : 1;          // distance between the outputs of one lane
    const int n_out = m >= 4 ? 4 : (int)m;        // outputs per lane (m = 1 or 2: a single lane)
    Quad2 a;
    a.p00 = a.p01 = a.p10 = a.q0 = a.q1 = a.t0 = a.t2 = a.t3 = ef_zero();
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < step; j += (u64)gridDim.x * 256) {
        EF fo[4], wo[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t >= n_out) {
                fo[t] = wo[t] = ef_zero();
                continue;
            }
            const u64 x = j + t * step;
            EF in[4];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int k = 0; k < 5; k++) in[s].v[k] = W[(u64)k * plane + x + s * m];
            // bit order: s = 2 b0 + b1 with b0 the variable of r0 (distance 2m), b1 that of r1 (distance m)
            wo[t] = ef_add(ef_add(in[0], ef_mul(r0, ef_sub(in[2], in[0]))),
                           ef_mul(r1, ef_sub(ef_add(in[1], ef_mul(r0, ef_sub(in[3], in[1]))), ef_add(in[0], ef_mul(r0, ef_sub(in[2], in[0]))))));
            if (F_BASE) {
                u32 b[4];
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = f[x + s * m];
                EF y0 = ef_mul_base(r0, sub(b[2], b[0])), y1 = ef_mul_base(r0, sub(b[3], b[1]));
                y0.v[0] = add(y0.v[0], b[0]);
                y1.v[0] = add(y1.v[0], b[1]);
                fo[t] = ef_add(y0, ef_mul(r1, ef_sub(y1, y0)));
            } else {
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int k = 0; k < 5; k++) in[s].v[k] = f[(u64)k * plane + x + s * m];
                const EF y0 = ef_add(in[0], ef_mul(r0, ef_sub(in[2], in[0]))), y1 = ef_add(in[1], ef_mul(r0, ef_sub(in[3], in[1])));
                fo[t] = ef_add(y0, ef_mul(r1, ef_sub(y1, y0)));
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                f_out[(u64)k * m + x] = fo[t].v[k];
                W_out[(u64)k * m + x] = wo[t].v[k];
            }
            // one output at a time: without the fence the ~100 loads (and their 64-bit addresses) of all four outputs are
            // requested up front — 256 VGPRs, 2 waves per SIMD, 2.7 TB/s; 24 loads in flight per lane are plenty at 4 waves
            asm volatile("" ::: "memory");
        }
        if (SUMS == 2) {
            quad_accumulate(fo, wo, a);
        } else if (SUMS == 1) {  // m = 2 (one pair) or larger: pairs (x, x + m/2) = outputs (0, 2) and (1, 3) of the lane when m >= 4
            if (m >= 4) {
                a.p00 = ef_add(a.p00, ef_add(ef_mul(fo[0], wo[0]), ef_mul(fo[1], wo[1])));
                a.p01 = ef_add(a.p01, ef_add(ef_mul(ef_sub(fo[2], fo[0]), ef_sub(wo[2], wo[0])), ef_mul(ef_sub(fo[3], fo[1]), ef_sub(wo[3], wo[1]))));
            } else {
                a.p00 = ef_add(a.p00, ef_mul(fo[0], wo[0]));
                a.p01 = ef_add(a.p01, ef_mul(ef_sub(fo[1], fo[0]), ef_sub(wo[1], wo[0])));
            }
        }
    }
    if (SUMS) finish40(a, red, acc, done_counter, final_out, seq);
}

// =====================================================================================================
// PoW: candidates base .. base + n; result = min hit (or 0xffffffff)
// =====================================================================================================
struct PowArgs {
    u32 pre[16];  // poseidon16_pow_base(capacity): the candidate-independent part of the first full round
    u32 base, n, mask, r2;
};
// The block that finishes last hands the result to the host (pinned buffer + sequence flag) and re-arms the device word:
// no separate publishing launch.
__global__ __launch_bounds__(256) void k_pow_grind(PowArgs a, u32* __restrict__ result, u32* __restrict__ done_counter,
                                                   u32* __restrict__ h_res, u32 seq) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const u32 w = a.base + i;
    if (i < a.n && w < P) {
        const u32 s8 = poseidon16_pow_word8(a.pre, mul(w, a.r2));  // (Montgomery form of the canonical candidate)
        if ((from_monty(s8) & a.mask) == 0) atomicMin(result, w);
    }
    lm_wait_stores();  // the atomicMin of this wave has been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lm_ticket(done_counter) == gridDim.x - 1) {
            lm_store_agent(done_counter, 0);
            lm_store_system(h_res, atomicExch(result, 0xffffffffu));
            lm_wait_stores();
            lm_publish_flag(h_res, seq);
        }
    }
}

// The same search with one candidate per DPP quad (poseidon16_quad.h): a ~2.2 k instruction chain on four times the lanes.
struct PowArgsQ {
    u32 cap[8];
    u32 base, n, mask, r2;
};
__global__ __launch_bounds__(256) void k_pow_grind_quad(PowArgsQ a, u32* __restrict__ result, u32* __restrict__ done_counter,
                                                        u32* __restrict__ h_res, u32 seq, const u32* __restrict__ tab) {
    __shared__ u32 lds[QUAD_TAB_WORDS];
    quad_load_table(lds, tab);
    __syncthreads();
    const u32 i = (blockIdx.x * 256 + threadIdx.x) >> 2, q = threadIdx.x & 3;
    const u32 w = a.base + i;
    if (i < a.n && w < P) {  // (whole quads: i is the same in the four lanes)
        u32 s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = q == 0 ? a.cap[k] : q == 1 ? a.cap[4 + k] : 0u;
        if (q == 2) s[0] = mul(w, a.r2);  // Montgomery form of the canonical candidate: state word 8
        quad_permute(s, lds + q * QUAD_STRIDE);
        if (q == 2 && (from_monty(s[0]) & a.mask) == 0) atomicMin(result, w);
    }
    lm_wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lm_ticket(done_counter) == gridDim.x - 1) {
            lm_store_agent(done_counter, 0);
            lm_store_system(h_res, atomicExch(result, 0xffffffffu));
            lm_wait_stores();
            lm_publish_flag(h_res, seq);
        }
    }
}

// init: W holds garbage on entry and must equal the sum on exit
static int weights_impl(lm_ctx* ctx, uint32_t* d_W, uint32_t n_vars, const lm_weight_item* items, uint32_t n_items,
                        const uint32_t* points, uint64_t n_point_coords, const uint32_t* scalars, bool init) {
    LM_REQUIRE(ctx && d_W && n_vars <= 40);
    const u64 plane = 1ull << n_vars;
    if (n_items == 0) {
        if (init) LM_HIP(hipMemsetAsync(d_W, 0, 5 * plane * 4, ctx->stream));
        return LM_OK;
    }
    LM_REQUIRE(n_vars <= 36);
    LM_REQUIRE(items && scalars && (points || n_point_coords == 0));
    // order items by (inner_n descending, offset): groups (same region) are contiguous, and all groups of one inner_n —
    // which are pairwise disjoint aligned blocks — form one launch.  Regions of different sizes may nest, so they must not
    // be updated by the same launch (read-modify-write of W).  Largest first: a whole-domain group (the OOD constraints of
    // combine_statement) then initialises W with plain stores instead of a memset plus a read-modify-write.
    std::vector<u32> order(n_items);
    std::vector<uint8_t> fast(n_items);
    for (u32 i = 0; i < n_items; i++) {
        order[i] = i;
        const lm_weight_item& s = items[i];
        LM_REQUIRE(s.inner_n <= n_vars && s.inner_n <= 36 && s.point_offset + s.inner_n <= n_point_coords);
        const u32 k_lo = s.inner_n < W_KLO ? s.inner_n : W_KLO;
        bool f = !s.is_next;  // fast: eq weight whose last k_lo coordinates are base-field elements
        for (u32 c = s.inner_n - k_lo; c < s.inner_n && f; c++) {
            const u32* e = points + (s.point_offset + c) * 5;
            if (e[1] | e[2] | e[3] | e[4]) f = false;
        }
        fast[i] = f;
    }
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) {
        if (items[a].inner_n != items[b].inner_n) return items[a].inner_n > items[b].inner_n;
        if (items[a].offset != items[b].offset) return items[a].offset < items[b].offset;
        return fast[a] > fast[b];  // fast items first within a group
    });
    std::vector<WItem> hit(n_items);
    std::vector<WGroup> hgr;
    std::vector<u32> hsc((u64)n_items * 5);
    u64 arena_words = 0;
    u32 max_table = 1;
    for (u32 k = 0; k < n_items; k++) {
        const lm_weight_item& s = items[order[k]];
        LM_REQUIRE(s.offset + (1ull << s.inner_n) <= plane && (s.offset & ((1ull << s.inner_n) - 1)) == 0);
        const u32 k_lo = s.inner_n < W_KLO ? s.inner_n : W_KLO;
        WItem& w = hit[k];
        w.inner_n = s.inner_n;
        w.is_next = s.is_next ? 1 : 0;
        w.point_off = s.point_offset;
        w.lo_base = fast[order[k]];
        w.pad = 0;
        w.thi_off = arena_words;
        arena_words += 5ull << (s.inner_n - k_lo);
        w.tlo_off = arena_words;
        arena_words += 5ull << k_lo;
        max_table = std::max(max_table, std::max(1u << (s.inner_n - k_lo), 1u << k_lo));
        memcpy(&hsc[(u64)k * 5], scalars + (u64)order[k] * 5, 20);
        if (hgr.empty() || hgr.back().offset != s.offset || hgr.back().inner_n != s.inner_n) {
            WGroup g;
            g.offset = s.offset;
            g.inner_n = s.inner_n;
            g.item_begin = k;
            g.item_end = k + 1;
            g.fast_end = k;
            g.chunk_begin = 0;
            hgr.push_back(g);
        } else {
            hgr.back().item_end = k + 1;
        }
        if (w.lo_base) hgr.back().fast_end = k + 1;
    }
    // init: one launch over the whole domain stores (whole-domain group) + (top-level region of the chunk); only regions
    // nested inside those, or smaller than a chunk, still need read-modify-write launches afterwards.
    std::vector<WGroup> merged, rest;
    if (init) {
        WGroup whole;
        whole.offset = 0;
        whole.inner_n = n_vars;
        whole.chunk_begin = 0;
        whole.item_begin = whole.item_end = whole.fast_end = 0;
        size_t first = 0;
        if (hgr[0].inner_n == n_vars) whole = hgr[first++];
        merged.push_back(whole);
        for (size_t i = first; i < hgr.size(); i++) {
            const WGroup& g = hgr[i];
            bool top = g.inner_n >= W_KLO;
            for (size_t m = 1; m < merged.size() && top; m++)  // hgr is sorted largest first: only earlier ones can contain g
                if (merged[m].offset <= g.offset && g.offset < merged[m].offset + (1ull << merged[m].inner_n)) top = false;
            (top ? merged : rest).push_back(g);
        }
        std::sort(merged.begin() + 1, merged.end(), [](const WGroup& a, const WGroup& b) { return a.offset < b.offset; });
    } else {
        rest = hgr;
    }
    u64 chunks = 0;
    for (size_t i = 0; i < rest.size(); i++) {  // chunk numbering restarts per launch (= per inner_n)
        if (i && rest[i - 1].inner_n != rest[i].inner_n) chunks = 0;
        rest[i].chunk_begin = chunks;
        chunks += ((1ull << rest[i].inner_n) + W_CHUNK - 1) / W_CHUNK;
        LM_REQUIRE(chunks < (1ull << 31));
    }
    LM_REQUIRE(((plane + W_CHUNK - 1) / W_CHUNK) < (1ull << 31));
    // scratch layout (words): items | merged groups | rest groups | scalars | points | arena
    const u64 w_items = (sizeof(WItem) * n_items + 3) / 4, w_merged = (sizeof(WGroup) * merged.size() + 3) / 4,
              w_rest = (sizeof(WGroup) * rest.size() + 3) / 4;
    const u64 w_sc = (u64)n_items * 5, w_pts = n_point_coords * 5;
    auto al = [](u64 x) { return (x + 15) & ~15ull; };
    // workgroups of k_weight_tables per item (prefix sums)
    std::vector<u32> blk_pre(n_items + 1, 0);
    for (u32 k = 0; k < n_items; k++) {
        const u32 k_lo = hit[k].inner_n < W_KLO ? hit[k].inner_n : W_KLO;
        const u32 longer = std::max(1u << (hit[k].inner_n - k_lo), 1u << k_lo);
        blk_pre[k + 1] = blk_pre[k] + (longer + 255) / 256;
    }
    const u64 o_items = 0, o_merged = al(o_items + w_items), o_rest = al(o_merged + w_merged), o_sc = al(o_rest + w_rest),
              o_pts = al(o_sc + w_sc), o_blk = al(o_pts + w_pts), o_arena = al(o_blk + n_items + 1);
    u32* s;
    int rc = lm_scratch(ctx, o_arena + arena_words, &s);
    if (rc) return rc;
    {
        // one host image of all tables (items | merged | rest | scalars | points) in the pinned staging ring, one copy command
        void* img;
        if ((rc = lm_stage_alloc(ctx, o_arena * 4, &img))) return rc;
        std::vector<u32> pageable;
        if (!img) {
            pageable.resize(o_arena);
            img = pageable.data();
        }
        u32* im = static_cast<u32*>(img);
        memcpy(im + o_items, hit.data(), sizeof(WItem) * n_items);
        if (!merged.empty()) memcpy(im + o_merged, merged.data(), sizeof(WGroup) * merged.size());
        if (!rest.empty()) memcpy(im + o_rest, rest.data(), sizeof(WGroup) * rest.size());
        memcpy(im + o_sc, hsc.data(), w_sc * 4);
        if (w_pts) memcpy(im + o_pts, points, w_pts * 4);
        memcpy(im + o_blk, blk_pre.data(), (n_items + 1) * 4ull);
        LM_HIP(hipMemcpyAsync(s, im, o_arena * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!pageable.empty()) LM_HIP(hipStreamSynchronize(ctx->stream));
    }
    (void)max_table;
    LM_LAUNCH(ctx, k_weight_tables, dim3(blk_pre[n_items], 2), dim3(256), 0, (const WItem*)(s + o_items), s + o_pts, s + o_sc, s + o_arena,
              (const u32*)(s + o_blk), n_items);
    const bool dbg = getenv("LM_DEBUG_WEIGHTS") != nullptr;
    if (init) {
        if (dbg) fprintf(stderr, "# weights init launch: whole-domain items=%u, top-level regions=%zu\n",
                         merged[0].item_end - merged[0].item_begin, merged.size() - 1);
        LM_LAUNCH(ctx, k_weights_init, dim3((unsigned)((plane + W_CHUNK - 1) / W_CHUNK)), dim3(256), 0, d_W, plane,
                  (const WGroup*)(s + o_merged), (u32)merged.size(), (const WItem*)(s + o_items), s + o_arena);
        LM_PROF_BYTES(ctx, k_weights_init, 20ull * plane);  // W written exactly once (5 planes)
    }
    for (size_t g0 = 0; g0 < rest.size();) {
        size_t g1 = g0;
        while (g1 < rest.size() && rest[g1].inner_n == rest[g0].inner_n) g1++;
        const WGroup& last = rest[g1 - 1];
        const u64 n_chunks = last.chunk_begin + ((1ull << last.inner_n) + W_CHUNK - 1) / W_CHUNK;
        if (dbg) fprintf(stderr, "# weights rmw launch inner_n=%u groups=%zu chunks=%llu\n", last.inner_n, g1 - g0,
                         (unsigned long long)n_chunks);
        LM_LAUNCH(ctx, k_weights_accumulate, dim3((unsigned)n_chunks), dim3(256), 0, d_W, plane,
                  (const WGroup*)(s + o_rest) + g0, (u32)(g1 - g0), (const WItem*)(s + o_items), s + o_arena);
        g0 = g1;
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" {

int lm_weights_accumulate(lm_ctx* ctx, uint32_t* d_W, uint32_t n_vars, const lm_weight_item* items, uint32_t n_items,
                          const uint32_t* points, uint64_t n_point_coords, const uint32_t* scalars) {
    return weights_impl(ctx, d_W, n_vars, items, n_items, points, n_point_coords, scalars, false);
}
int lm_weights_init(lm_ctx* ctx, uint32_t* d_W, uint32_t n_vars, const lm_weight_item* items, uint32_t n_items,
                    const uint32_t* points, uint64_t n_point_coords, const uint32_t* scalars) {
    return weights_impl(ctx, d_W, n_vars, items, n_items, points, n_point_coords, scalars, true);
}

int lm_prod_round(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars,
                  uint32_t out_c0_c2[10]) {
    LM_REQUIRE(ctx && d_f && d_W && out_c0_c2 && n_vars >= 1 && n_vars <= 40);
    const u64 half = 1ull << (n_vars - 1);
    u32 blocks = half <= 512 ? 1 : (u32)std::min<u64>((half + 255) / 256, 2048);
    const u32 seq = ++ctx->res_seq;
    u32* s;
    int rc = lm_scratch(ctx, (u64)blocks * 10 + 16, &s);
    if (rc) return rc;
    u32* d_out = s + (u64)blocks * 10;
    if (f_is_ext)
        LM_LAUNCH(ctx, k_prod_round_ext, dim3(blocks), dim3(256), 0, d_f, d_W, half, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq);
    else
        LM_LAUNCH(ctx, k_prod_round_base, dim3(blocks), dim3(256), 0, d_f, d_W, half, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq);
    (void)d_out;
    LM_HIP(hipGetLastError());
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(out_c0_c2, ctx->h_res, 40);
    return LM_OK;
}

int lm_fold(lm_ctx* ctx, const uint32_t* d_in, int in_is_ext, uint32_t n_vars, const uint32_t r[LM_EF_DIM],
            uint32_t* d_out) {
    LM_REQUIRE(ctx && d_in && d_out && r && n_vars >= 1 && n_vars <= 40);
    const u64 half = 1ull << (n_vars - 1);
    u32 blocks = (u32)std::min<u64>((half + 255) / 256, 4096);
    EF rr;
    memcpy(rr.v, r, 20);
    if (in_is_ext)
        LM_LAUNCH(ctx, k_fold_ext, dim3(blocks), dim3(256), 0, d_in, half, rr, d_out);
    else
        LM_LAUNCH(ctx, k_fold_base, dim3(blocks), dim3(256), 0, d_in, half, rr, d_out);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

int lm_fold_round(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars, const uint32_t r[LM_EF_DIM],
                  uint32_t* d_f_out, uint32_t* d_W_out, uint32_t out_c0_c2[10]) {
    LM_REQUIRE(ctx && d_f && d_W && d_f_out && d_W_out && r && out_c0_c2 && n_vars >= 2 && n_vars <= 40);
    const u64 half = 1ull << (n_vars - 1), quarter = half >> 1;
    const u32 blocks = quarter <= 512 ? 1 : (u32)std::min<u64>((quarter + 255) / 256, 4096);
    const u32 seq = ++ctx->res_seq;
    u32* s;
    int rc = lm_scratch(ctx, (u64)blocks * 10 + 16, &s);
    if (rc) return rc;
    EF rr;
    memcpy(rr.v, r, 20);
    if (f_is_ext)
        LM_LAUNCH(ctx, (k_fold_round<false>), dim3(blocks), dim3(256), 0, d_f, d_W, half, rr, d_f_out, d_W_out, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq);
    else
        LM_LAUNCH(ctx, (k_fold_round<true>), dim3(blocks), dim3(256), 0, d_f, d_W, half, rr, d_f_out, d_W_out, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq);
    LM_HIP(hipGetLastError());
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(out_c0_c2, ctx->h_res, 40);
    return LM_OK;
}

int lm_prod_round2(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars, uint32_t out_sums[40]) {
    LM_REQUIRE(ctx && d_f && d_W && out_sums && n_vars >= 2 && n_vars <= 40);
    const u64 quarter = 1ull << (n_vars - 2);
    // 512 workgroups: the kernel is a grid-stride loop that ends in 40 accumulator atomics per workgroup (lm_grid_sum); measured on
    // the 1.6 GB pass — 256: 0.42 ms, 384: 0.42, 512: 0.40, 768: 0.42, 1024 / 2048 / 4096: 0.44
    const u32 blocks = (u32)std::min<u64>((quarter + 255) / 256, 512);
    const u32 seq = ++ctx->res_seq;
    LM_PROF_BYTES(ctx, k_prod_round2, (4 * quarter) * (f_is_ext ? 40ull : 24ull));  // every f and W value once
    if (f_is_ext)
        LM_LAUNCH(ctx, (k_prod_round2<false>), dim3(blocks), dim3(256), 0, d_f, d_W, quarter, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq);
    else
        LM_LAUNCH(ctx, (k_prod_round2<true>), dim3(blocks), dim3(256), 0, d_f, d_W, quarter, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq);
    LM_HIP(hipGetLastError());
    int rc;
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(out_sums, ctx->h_res, PR2_WORDS * 4);
    return LM_OK;
}

int lm_fold2_round(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars, const uint32_t r0[LM_EF_DIM],
                   const uint32_t r1[LM_EF_DIM], uint32_t* d_f_out, uint32_t* d_W_out, int sums, uint32_t* out_sums) {
    LM_REQUIRE(ctx && d_f && d_W && d_f_out && d_W_out && r0 && r1 && n_vars >= 2 && n_vars <= 40 && sums >= 0 && sums <= 2);
    LM_REQUIRE(sums == 0 || out_sums);
    const u64 m = 1ull << (n_vars - 2);
    LM_REQUIRE(sums < 2 || m >= 4);
    LM_REQUIRE(sums < 1 || m >= 2);
    const u64 lanes = m >= 4 ? m / 4 : 1;
    const u32 blocks = (u32)std::min<u64>((lanes + 255) / 256, 2048);  // (512 .. 4096 workgroups: within 3 % of each other)
    EF a, b;
    memcpy(a.v, r0, 20);
    memcpy(b.v, r1, 20);
    const u32 seq = sums ? ++ctx->res_seq : 0;
    LM_PROF_BYTES(ctx, k_fold2_round, (4 * m) * (f_is_ext ? 40ull : 24ull) + m * 40ull);  // reads f and W once, writes both folded tables
#define F2(FB, S) LM_LAUNCH(ctx, (k_fold2_round<FB, S>), dim3(blocks), dim3(256), 0, d_f, d_W, m, a, b, d_f_out, d_W_out, ctx->d_acc, ctx->d_sync + 1, ctx->h_res, seq)
    if (f_is_ext) {
        if (sums == 2)
            F2(false, 2);
        else if (sums == 1)
            F2(false, 1);
        else
            F2(false, 0);
    } else {
        if (sums == 2)
            F2(true, 2);
        else if (sums == 1)
            F2(true, 1);
        else
            F2(true, 0);
    }
#undef F2
    LM_HIP(hipGetLastError());
    if (!sums) return LM_OK;
    int rc;
    if ((rc = lm_wait_result(ctx, seq))) return rc;
    memcpy(out_sums, ctx->h_res, (sums == 2 ? PR2_WORDS : 10) * 4);
    return LM_OK;
}

int lm_pow_grind(lm_ctx* ctx, const uint32_t capacity[8], uint32_t bits, uint32_t* witness) {
    LM_REQUIRE(ctx && capacity && witness && bits < 31);
    if (bits == 0) {
        *witness = 0;
        return LM_OK;
    }
    // Two kernels (measured on MI355X, kernel time per launch):
    //   one candidate per lane   2^17 candidates: 36 us (2 waves per SIMD of a ~6 k instruction chain: latency bound; 5.7 G/s asymptotically)
    //   one candidate per quad   2^15: 19 us, 2^16: 25 us, 2^17: 43 us (a ~2.4 k chain, 1.6x the instructions: 3.4 G/s asymptotically)
    // The expected work is 2^bits candidates and a failed batch costs another round trip (~8 us), so batch ~ 2 x 2^bits:
    // searches of <= 14 bits (half of the 30 per proof) take the quad kernel with 2^15 candidates, the others the lane kernel.
    // LM_POW_SINGLE_LANE=1 / LM_POW_QUAD=1 force one kernel, LM_POW_BATCH_LOG the batch (experiments).
    static const bool force_lane = getenv("LM_POW_SINGLE_LANE") != nullptr, force_quad = getenv("LM_POW_QUAD") != nullptr;
    static const char* batch_env = getenv("LM_POW_BATCH_LOG");
    const bool single_lane = force_lane || (!force_quad && bits > 14);
    PowArgs a;
    PowArgsQ aq;
    poseidon16_pow_base(capacity, a.pre);
    memcpy(aq.cap, capacity, 32);
    a.mask = aq.mask = (1u << bits) - 1;
    a.r2 = aq.r2 = to_monty(to_monty(1));  // 2^64 mod p
    u64 batch = single_lane ? std::max<u64>(1ull << 17, std::min<u64>(2ull << bits, 1ull << 22))
                            : std::min<u64>(std::max<u64>(2ull << bits, 1ull << 13), 1ull << 17);
    if (batch_env) batch = 1ull << atoi(batch_env);
    for (u64 base = 0; base < P; base += batch) {
        a.base = aq.base = (u32)base;
        a.n = aq.n = (u32)std::min<u64>(batch, (u64)P - base);
        const u32 seq = ++ctx->res_seq;
        if (single_lane)
            LM_LAUNCH(ctx, k_pow_grind, dim3((a.n + 255) / 256), dim3(256), 0, a, ctx->d_sync, ctx->d_sync + 1, ctx->h_res, seq);
        else
            LM_LAUNCH(ctx, k_pow_grind_quad, dim3((unsigned)(((u64)aq.n * 4 + 255) / 256)), dim3(256), 0, aq, ctx->d_sync, ctx->d_sync + 1,
                      ctx->h_res, seq, (const u32*)ctx->d_quad);
        LM_HIP(hipGetLastError());
        int rc = lm_wait_result(ctx, seq);
        if (rc) return rc;
        const u32 res = ctx->h_res[0];
        if (res != 0xffffffffu) {
            *witness = to_monty(res);
            return LM_OK;
        }
    }
    lm_set_error("lm_pow_grind: no witness");
    return LM_E_INVALID;
}

}  // extern "C"
