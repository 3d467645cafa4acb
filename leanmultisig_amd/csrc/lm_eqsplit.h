// Shared device helpers: split eq tables for "eq over a shrinking prefix of a point" (used by the LSB-first sumchecks of
// the logup GKR and of the AIR tables) and EF block reductions.
#pragma once
#include <algorithm>
#include <vector>
#include "lm_common.h"

// eq(point[0..p), j) for every prefix length p <= P1 is served from small tables:
//   p <= H : one table over the first p coordinates
//   p >  H : T_hi over the first H coordinates  x  T_lo over coordinates [H, p)
// with H = ceil(P1 / 2).  All tables of a point are built by one launch.
// The point and the table layout travel as kernel arguments (no host-to-device copy, no synchronisation per sumcheck):
// table d <= H: coordinates [0, d), offset 5 (2^d - 1); table d > H: coordinates [H, d), after the hi tables.
static constexpr kb::u32 EQ_MAX_COORDS = 32;
struct EqPointArg {
    kb::u32 v[EQ_MAX_COORDS * 5];
    kb::u32 H, P1;
};
__host__ __device__ inline kb::u64 eq_table_offset(kb::u32 d, kb::u32 H) {
    if (d <= H) return 5ull * ((1ull << d) - 1);
    return 5ull * ((1ull << (H + 1)) - 1) + 5ull * ((1ull << (d - H)) - 2);  // + sum_{p=H+1}^{d-1} 2^(p-H)
}
template <int UNUSED>
__global__ __launch_bounds__(256) void k_prefix_eq_tables(EqPointArg pt, kb::u32* __restrict__ arena) {
    using namespace kb;
    const u32 d = blockIdx.y;  // 0 .. P1
    const u32 c0 = d <= pt.H ? 0 : pt.H, nb = d <= pt.H ? d : d - pt.H;
    const u64 off = eq_table_offset(d, pt.H);
    const u32 len = 1u << nb;
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    EF acc = ef_one();
    for (u32 j = 0; j < nb; j++) {
        EF p;
#pragma unroll
        for (int k = 0; k < 5; k++) p.v[k] = pt.v[(c0 + j) * 5 + k];
        u32 bit = (i >> (nb - 1 - j)) & 1;
        acc = ef_mul(acc, bit ? p : ef_sub(ef_one(), p));
    }
#pragma unroll
    for (int k = 0; k < 5; k++) arena[off + (u64)k * len + i] = acc.v[k];
}

struct EqSplit {
    const kb::u32* th;  // hi table (SoA, len_hi entries)
    const kb::u32* tl;  // lo table (SoA, 2^log_lo entries) or nullptr
    kb::u32 len_hi, log_lo;
};
__device__ __forceinline__ kb::EF eq_split_at(const EqSplit& e, kb::u64 j) {
    using namespace kb;
    EF a;
    const u64 jh = j >> e.log_lo;
#pragma unroll
    for (int k = 0; k < 5; k++) a.v[k] = e.th[(u64)k * e.len_hi + jh];
    if (e.tl) {
        EF b;
        const u32 len_lo = 1u << e.log_lo;
        const u32 jl = (u32)j & (len_lo - 1);
#pragma unroll
        for (int k = 0; k < 5; k++) b.v[k] = e.tl[(u64)k * len_lo + jl];
        a = ef_mul(a, b);
    }
    return a;
}

// Host side: table bookkeeping for one point of P1 + 1 coordinates (prefix lengths 0..P1 are served).
struct PrefixEqTables {
    kb::u32* d_buf = nullptr;  // device buffer owned by the user of this struct
    kb::u64 buf_words = 0;
    kb::u32 H = 0;
    std::vector<kb::u64> th_off, tl_off;

    static kb::u64 words_needed(kb::u32 max_coords) { return (20ull << ((max_coords + 1) / 2 + 1)) + 2048 + 5ull * max_coords; }

    // point: host, n_coords x 5 words; tables for prefixes of the first P1 = n_coords - 1 coordinates
    int build(lm_ctx* ctx, const kb::u32* point, kb::u32 n_coords) {
        using namespace kb;
        LM_REQUIRE(d_buf && n_coords >= 1 && n_coords <= EQ_MAX_COORDS);
        const u32 P1 = n_coords - 1;
        H = (P1 + 1) / 2;
        th_off.assign(H + 1, 0);
        tl_off.assign(P1 + 1, 0);
        u32 max_len = 1;
        for (u32 q = 0; q <= H; q++) {
            th_off[q] = eq_table_offset(q, H);
            max_len = std::max(max_len, 1u << q);
        }
        for (u32 p = H + 1; p <= P1; p++) {
            tl_off[p] = eq_table_offset(p, H);
            max_len = std::max(max_len, 1u << (p - H));
        }
        LM_REQUIRE(eq_table_offset(P1 + 1, H) <= buf_words || P1 + 1 <= H);
        EqPointArg a;
        memcpy(a.v, point, (size_t)n_coords * 20);
        a.H = H;
        a.P1 = P1;
        LM_LAUNCH(ctx, k_prefix_eq_tables<0>, dim3((max_len + 255) / 256, P1 + 1), dim3(256), 0, a, d_buf);
        LM_HIP(hipGetLastError());
        return LM_OK;
    }
    EqSplit at(kb::u32 p) const {
        EqSplit e;
        if (p <= H) {
            e.th = d_buf + th_off[p];
            e.len_hi = 1u << p;
            e.tl = nullptr;
            e.log_lo = 0;
        } else {
            e.th = d_buf + th_off[H];
            e.len_hi = 1u << H;
            e.tl = d_buf + tl_off[p];
            e.log_lo = p - H;
        }
        return e;
    }
};

__device__ __forceinline__ kb::u32 wave_sum_u32(kb::u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, (kb::u32)__shfl_down(v, off, 64));
    return v;
}
