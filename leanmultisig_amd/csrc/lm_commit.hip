// WHIR commitment on gfx950: LDE (gather/replicate + evaluation-domain radix-2 NTT), Poseidon1-16 leaf sponge,
// Merkle levels, batched openings.
//
// HBM layout: the LDE matrix is COLUMN-major (one column = h contiguous words) — column c of the reference's
// row-major matrix (crates/whir/src/utils.rs:128-150) is the c-th contiguous slice of the committed polynomial, so
// the gather is a replicate-on-load, every NTT pass streams whole columns, and the leaf sponge (one row per lane)
// reads 64 consecutive rows of a column per wave instruction: fully coalesced.  An EF matrix of C columns is stored
// as 5*C base columns in plane-major order (k*C + c); the reference's leaf word 5c+k maps to that column.
#include <memory>
#include <vector>
#include "lm_common.h"
#include "poseidon16_coop.h"
#include "poseidon16_quad.h"

using namespace kb;

// =====================================================================================================
// NTT.  Layer l (1-based) of the reference (crates/whir/src/dft.rs:79-144,548-568): blocks of 2^l rows, j < 2^(l-1):
//   a = v[j], b = v[j + 2^(l-1)], d = (b - a) * w_{2^l}^j, v[j] = a + d, v[j + 2^(l-1)] = a - d.
// One pass executes layers S+1 .. S+K on an LDS tile of 2^K "hi" values x 2^m consecutive "lo" values:
//   element e = blk * 2^(S+K) + hi * 2^S + lo.
// Twiddle of layer S+q+1 for (hi, lo):  w_{2^(q+1)}^(hi mod 2^q) * w_{2^(S+q+1)}^lo ; the first factor comes from the
// small layered table (LDS), the second depends only on (q, lo) and is staged once per workgroup.
// =====================================================================================================
struct NttArgs {
    u32 log_h, S, K, m;
    u32 first;      // 1: read (replicated) input, 0: in place
    u32 log_rate;   // replicate factor of the first pass
    u64 in_col_len; // input words per column (first pass)
    u32 cols_per_plane;   // input column c lives at (c / cols_per_plane) * in_plane_stride + (c % cols_per_plane) * in_col_len
    u64 in_plane_stride;  // (EF polynomials: 5 SoA planes of 2^n_vars words; the output matrix is plane-major contiguous)
};

__global__ __launch_bounds__(256) void k_ntt_pass(const u32* __restrict__ in, u32* __restrict__ out,
                                                  const u32* __restrict__ tw_big, const u32* __restrict__ tw_small,
                                                  NttArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 K = a.K, m = a.m, S = a.S;
    const u32 tile = 1u << (K + m);
    u32* data = lds;
    u32* tws = lds + tile;              // 2^K words
    u32* twlo = tws + (1u << K);        // K * 2^m words (S > 0 only)
    const u32 tid = threadIdx.x;
    const u64 h = 1ull << a.log_h;
    const u64 col = blockIdx.y;
    const u32 t = blockIdx.x;
    const u32 lo_tiles_log = S - m;  // S >= m
    const u64 blk = t >> lo_tiles_log;
    const u32 lo_tile = t & ((1u << lo_tiles_log) - 1);
    const u64 base = (blk << (S + K)) + ((u64)lo_tile << m);

    for (u32 x = tid; x < (1u << K); x += 256) tws[x] = tw_small[x];
    if (S > 0) {
        for (u32 x = tid; x < (K << m); x += 256) {
            u32 q = x >> m, lo_l = x & ((1u << m) - 1);
            u32 l = S + q + 1;
            u64 lo = ((u64)lo_tile << m) + lo_l;
            twlo[x] = tw_big[lo << (LM_TW_LOG - l)];
        }
    }
    if (a.first) {
        const u32* src = in + (col / a.cols_per_plane) * a.in_plane_stride + (col % a.cols_per_plane) * a.in_col_len;
        for (u32 x = tid; x < tile; x += 256) {
            u32 hi = x >> m, lo_l = x & ((1u << m) - 1);
            u64 e = base + ((u64)hi << S) + lo_l;
            data[x] = src[e >> a.log_rate];
        }
    } else {
        const u32* src = out + col * h;
        for (u32 x = tid; x < tile; x += 256) {
            u32 hi = x >> m, lo_l = x & ((1u << m) - 1);
            u64 e = base + ((u64)hi << S) + lo_l;
            data[x] = src[e];
        }
    }
    // Layers in groups of three (radix 8 in registers: 8 LDS reads + 8 writes and one index computation per 12 butterflies),
    // the remainder as radix 4 / radix 2.  T(l, idx, lo) = twiddle of layer l for hi-index idx.
    auto tw_of = [&](u32 l, u32 idx, u32 lo_l) {
        u32 w = tws[(1u << l) + idx];
        if (S > 0) w = mul(w, twlo[(l << m) + lo_l]);
        return w;
    };
    auto bfly = [&](u32& a_, u32& b_, u32 w) {
        const u32 d = mul(sub(b_, a_), w);
        const u32 t = a_;
        a_ = add(t, d);
        b_ = sub(t, d);
    };
    u32 q = 0;
    for (; q + 3 <= K; q += 3) {
        __syncthreads();
        const u32 bp = q + m;
        for (u32 g = tid; g < (tile >> 3); g += 256) {
            const u32 x0 = ((g >> bp) << (bp + 3)) | (g & ((1u << bp) - 1));
            const u32 tq = (x0 >> m) & ((1u << q) - 1), lo_l = x0 & ((1u << m) - 1);
            u32 v[8];
            if (bp == 0) {  // 8 consecutive words
                const uint4 p0 = *reinterpret_cast<const uint4*>(data + x0), p1 = *reinterpret_cast<const uint4*>(data + x0 + 4);
                v[0] = p0.x, v[1] = p0.y, v[2] = p0.z, v[3] = p0.w, v[4] = p1.x, v[5] = p1.y, v[6] = p1.z, v[7] = p1.w;
            } else {
#pragma unroll
                for (u32 k = 0; k < 8; k++) v[k] = data[x0 + (k << bp)];
            }
            const u32 wa = tw_of(q, tq, lo_l);
            bfly(v[0], v[1], wa), bfly(v[2], v[3], wa), bfly(v[4], v[5], wa), bfly(v[6], v[7], wa);
            const u32 wb0 = tw_of(q + 1, tq, lo_l), wb1 = tw_of(q + 1, tq + (1u << q), lo_l);
            bfly(v[0], v[2], wb0), bfly(v[1], v[3], wb1), bfly(v[4], v[6], wb0), bfly(v[5], v[7], wb1);
#pragma unroll
            for (u32 k = 0; k < 4; k++) bfly(v[k], v[k + 4], tw_of(q + 2, tq + (k << q), lo_l));
            if (bp == 0) {
                *reinterpret_cast<uint4*>(data + x0) = make_uint4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<uint4*>(data + x0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
                for (u32 k = 0; k < 8; k++) data[x0 + (k << bp)] = v[k];
            }
        }
    }
    if (q + 2 == K) {
        __syncthreads();
        const u32 bp = q + m;
        for (u32 g = tid; g < (tile >> 2); g += 256) {
            const u32 x0 = ((g >> bp) << (bp + 2)) | (g & ((1u << bp) - 1));
            const u32 tq = (x0 >> m) & ((1u << q) - 1), lo_l = x0 & ((1u << m) - 1);
            u32 v[4];
#pragma unroll
            for (u32 k = 0; k < 4; k++) v[k] = data[x0 + (k << bp)];
            const u32 wa = tw_of(q, tq, lo_l);
            bfly(v[0], v[1], wa), bfly(v[2], v[3], wa);
            bfly(v[0], v[2], tw_of(q + 1, tq, lo_l)), bfly(v[1], v[3], tw_of(q + 1, tq + (1u << q), lo_l));
#pragma unroll
            for (u32 k = 0; k < 4; k++) data[x0 + (k << bp)] = v[k];
        }
        q += 2;
    }
    for (; q < K; q++) {
        __syncthreads();
        const u32 bp = q + m;
        for (u32 b = tid; b < (tile >> 1); b += 256) {
            const u32 x0 = ((b >> bp) << (bp + 1)) | (b & ((1u << bp) - 1));
            const u32 x1 = x0 | (1u << bp);
            const u32 w = tw_of(q, (x0 >> m) & ((1u << q) - 1), x0 & ((1u << m) - 1));
            u32 va = data[x0], vb = data[x1];
            bfly(va, vb, w);
            data[x0] = va;
            data[x1] = vb;
        }
    }
    __syncthreads();
    u32* dst = out + col * h;
    for (u32 x = tid; x < tile; x += 256) {
        u32 hi = x >> m, lo_l = x & ((1u << m) - 1);
        u64 e = base + ((u64)hi << S) + lo_l;
        dst[e] = data[x];
    }
}

// ---- register-resident radix-16 passes (round 4) ---------------------------------------------------------------------------------
// k_ntt_pass does three layers per LDS round trip with one index computation per 12 butterflies and, when S > 0, two Montgomery
// products per butterfly (46 lane-operations per butterfly measured).  The two kernels below are the same layers for the shapes the
// big matrices take (the first 12 layers of a column; layers S+1 .. S+8 behind them): a thread keeps SIXTEEN values in registers for
// four layers (32 butterflies), the values cross lanes through LDS once between such phases and enter / leave through registers:
//   k_ntt_first12   global -> registers (16 consecutive outputs of the replicated input: the first log_rate layers see equal pairs and
//                   are skipped, twiddles of index 0 are 1: 15 of the 32 butterflies of the first phase need no product) -> LDS ->
//                   registers (stride 16) -> LDS -> registers (stride 256) -> global (coalesced): 2 LDS writes + 2 reads per value
//                   instead of 5 + 5; twiddles straight from the layered table (L1 / scalar cache), no per-workgroup staging;
//   k_ntt_second8   tile = 256 block indices (hi) x 32 consecutive positions (lo): global -> registers (hi = 16 G + k) -> LDS ->
//                   registers (hi = t + 16 k) -> global.  The twiddle of layer S + q + 1 is w_{2^(q+1)}^(hi mod 2^q) * w_{2^(S+q+1)}^lo:
//                   the second factor depends on (q, lo) — one LDS word per lane and layer —, the first on the thread's 15 pair
//                   classes per phase: 15 products per 32 butterflies (11 in the first phase, where hi mod 2^q = 0 gives 1)
//                   instead of 32.
template <int S0>
__device__ __forceinline__ void ntt_phase16(u32 (&v)[16], const u32 (&w)[15], u32 skip_layers) {
    // four layers on 16 values: layer s pairs (i, i + 2^s), twiddle class i mod 2^s -> w[(1 << s) - 1 + class]
    static_for<0, 4>([&](auto SS) {
        constexpr int s = decltype(SS)::value;
        if ((u32)s < skip_layers) return;  // replicated input: both values of a pair are equal, the layer is the identity
        static_for<0, 16>([&](auto II) {
            constexpr int i = decltype(II)::value;
            if constexpr ((i >> s & 1) == 0) {
                constexpr int cls = i & ((1 << s) - 1);
                u32 d = sub(v[i + (1 << s)], v[i]);
                if constexpr (!(S0 == 0 && cls == 0)) d = mul(d, w[(1 << s) - 1 + cls]);  // (S0 == 0: class 0 is w^0 = 1)
                const u32 t = v[i];
                v[i] = add(t, d);
                v[i + (1 << s)] = sub(t, d);
            }
        });
    });
}
__device__ __forceinline__ u32 ntt_pad(u32 x) { return x + ((x >> 8) << 4); }  // 16 words of padding per 256: stride-256 columns spread over the banks

__global__ __launch_bounds__(256) void k_ntt_first12(const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ tw_small, NttArgs a) {
    __shared__ __attribute__((aligned(16))) u32 data[4096 + 256];
    const u32 g = threadIdx.x;
    const u64 col = blockIdx.y, h = 1ull << a.log_h;
    const u64 base = (u64)blockIdx.x << 12;
    const u32* src = in + (col / a.cols_per_plane) * a.in_plane_stride + (col % a.cols_per_plane) * a.in_col_len;
    u32 v[16], w[15];
    // phase 1: layers 1..4 on 16 consecutive outputs
    if (a.log_rate == 1) {
        const uint4* p = reinterpret_cast<const uint4*>(src + ((base + 16 * g) >> 1));
        const uint4 p0 = p[0], p1 = p[1];
        v[0] = v[1] = p0.x, v[2] = v[3] = p0.y, v[4] = v[5] = p0.z, v[6] = v[7] = p0.w;
        v[8] = v[9] = p1.x, v[10] = v[11] = p1.y, v[12] = v[13] = p1.z, v[14] = v[15] = p1.w;
    } else {
#pragma unroll
        for (u32 k = 0; k < 16; k++) v[k] = src[(base + 16 * g + k) >> a.log_rate];
    }
#pragma unroll
    for (u32 k = 0; k < 15; k++) w[k] = tw_small[k + 1];  // entry (1 << s) + j = w_{2^(s+1)}^j (uniform: scalar loads)
    ntt_phase16<0>(v, w, a.log_rate < 4 ? a.log_rate : 4);
    {
        uint4* d4 = reinterpret_cast<uint4*>(data + ntt_pad(16 * g));
        d4[0] = make_uint4(v[0], v[1], v[2], v[3]), d4[1] = make_uint4(v[4], v[5], v[6], v[7]);
        d4[2] = make_uint4(v[8], v[9], v[10], v[11]), d4[3] = make_uint4(v[12], v[13], v[14], v[15]);
    }
    // phase 2: layers 5..8, values x0 + 16 k, twiddle class of the thread tq = g & 15
    {
        const u32 tq = g & 15;
#pragma unroll
        for (u32 s = 0; s < 4; s++)
#pragma unroll
            for (u32 j = 0; j < (1u << s); j++) w[(1u << s) - 1 + j] = tw_small[(16u << s) + tq + 16 * j];
    }
    __syncthreads();
    {
        const u32 x0 = ((g >> 4) << 8) | (g & 15);
#pragma unroll
        for (u32 k = 0; k < 16; k++) v[k] = data[ntt_pad(x0 + 16 * k)];
        ntt_phase16<4>(v, w, 0);
#pragma unroll
        for (u32 k = 0; k < 16; k++) data[ntt_pad(x0 + 16 * k)] = v[k];
    }
    // phase 3: layers 9..12, values g + 256 k, twiddle class g
#pragma unroll
    for (u32 s = 0; s < 4; s++)
#pragma unroll
        for (u32 j = 0; j < (1u << s); j++) w[(1u << s) - 1 + j] = tw_small[(256u << s) + g + 256 * j];
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 16; k++) v[k] = data[g + 272 * k];  // ntt_pad(g + 256 k)
    ntt_phase16<8>(v, w, 0);
    u32* dst = out + col * h + base + g;
#pragma unroll
    for (u32 k = 0; k < 16; k++) dst[256 * k] = v[k];
}

// layers S + 1 .. S + 8 in place (S >= 5): workgroup = 512 threads, tile = 256 hi x 32 lo
__global__ __launch_bounds__(512) void k_ntt_second8(u32* __restrict__ out, const u32* __restrict__ tw_big, const u32* __restrict__ tw_small, NttArgs a) {
    __shared__ u32 data[8192];
    __shared__ u32 twlo[8 * 32];
    const u32 tid = threadIdx.x, lo_l = tid & 31, G = tid >> 5;
    const u32 S = a.S;
    const u64 h = 1ull << a.log_h;
    const u32 lo_tiles_log = S - 5;
    const u64 blk = blockIdx.x >> lo_tiles_log;
    const u32 lo_tile = blockIdx.x & ((1u << lo_tiles_log) - 1);
    const u64 base = (blk << (S + 8)) + ((u64)lo_tile << 5);
    u32* col = out + (u64)blockIdx.y * h + base + lo_l;
    u32 v[16], w[15];
#pragma unroll
    for (u32 k = 0; k < 16; k++) v[k] = col[(u64)(16 * G + k) << S];
    if (tid < 256) {  // w_{2^(S+q+1)}^lo for the 8 layers x 32 positions of the tile
        const u32 q = tid >> 5;
        const u64 lo = ((u64)lo_tile << 5) + lo_l;
        twlo[tid] = tw_big[lo << (LM_TW_LOG - (S + q + 1))];
    }
    __syncthreads();
    // phase A: layers q = 0..3 over k (hi = 16 G + k: hi mod 2^q = k mod 2^q, the same classes in every thread)
#pragma unroll
    for (u32 s = 0; s < 4; s++) {
        const u32 u = twlo[32 * s + lo_l];
        w[(1u << s) - 1] = u;  // class 0: w^0 = 1
#pragma unroll
        for (u32 j = 1; j < (1u << s); j++) w[(1u << s) - 1 + j] = mul(u, tw_small[(1u << s) + j]);
    }
    ntt_phase16<1>(v, w, 0);
#pragma unroll
    for (u32 k = 0; k < 16; k++) data[(16 * G + k) * 32 + lo_l] = v[k];
    // phase B: layers q = 4..7 over hi = G + 16 k: class of pair (q, j) = G + 16 j
#pragma unroll
    for (u32 s = 0; s < 4; s++) {
        const u32 u = twlo[32 * (4 + s) + lo_l];
#pragma unroll
        for (u32 j = 0; j < (1u << s); j++) w[(1u << s) - 1 + j] = mul(u, tw_small[(16u << s) + G + 16 * j]);
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 16; k++) v[k] = data[(G + 16 * k) * 32 + lo_l];
    ntt_phase16<5>(v, w, 0);
#pragma unroll
    for (u32 k = 0; k < 16; k++) col[(u64)(G + 16 * k) << S] = v[k];
}

// columns: n_planes x cols_per_plane inputs of in_col_len = h >> log_rate words (planes in_plane_stride apart) at d_in;
// output column-major (n_planes * cols_per_plane) x h.  All planes go through the same launches.
static int lde_columns(lm_ctx* ctx, const u32* d_in, u32* d_out, u32 cols_per_plane, u32 n_planes, u64 in_plane_stride, u32 log_h,
                       u32 log_rate) {
    const u32 n_cols = cols_per_plane * n_planes;
    LM_REQUIRE(log_h <= LM_TW_LOG);
    if (log_h == 0) {
        for (u32 k = 0; k < n_planes; k++)
            LM_HIP(hipMemcpyAsync(d_out + (u64)k * cols_per_plane, d_in + k * in_plane_stride, (u64)cols_per_plane * 4,
                                  hipMemcpyDeviceToDevice, ctx->stream));
        return LM_OK;
    }
    static const bool radix16 = getenv("LM_NTT_RADIX8") == nullptr;  // LM_NTT_RADIX8=1: the LDS radix-8 passes only (A/B measurements)
    const u32 K1 = log_h < 12 ? log_h : 12;
    const u64 in_col_len = (1ull << log_h) >> log_rate;
    u32 done = 0;
    {
        NttArgs a{log_h, 0, K1, 0, 1, log_rate, in_col_len, cols_per_plane, in_plane_stride};
        dim3 grid(1u << (log_h - K1), n_cols);
        // (the uint4 loads of the rate-1/2 path need 16-byte aligned columns)
        const bool aligned = log_rate != 1 || (((uintptr_t)d_in | (in_col_len * 4) | (in_plane_stride * 4)) & 15) == 0;
        if (radix16 && K1 == 12 && aligned) {
            LM_LAUNCH(ctx, k_ntt_first12, grid, dim3(256), 0, d_in, d_out, (const u32*)ctx->d_tw_small, a);
        } else {
            size_t sh = ((1u << K1) * 2) * 4;
            LM_LAUNCH(ctx, k_ntt_pass, grid, dim3(256), sh, d_in, d_out, ctx->d_tw, ctx->d_tw_small, a);
        }
        done = K1;
    }
    u32 rem = log_h - done;
    u32 n_pass = (rem + 8) / 9;
    for (u32 p = 0; p < n_pass; p++) {
        u32 K = (rem + (n_pass - p) - 1) / (n_pass - p);
        u32 m = 13 - K;
        if (m > done) m = done;
        NttArgs a{log_h, done, K, m, 0, 0, 0, cols_per_plane, in_plane_stride};
        if (radix16 && K == 8 && m == 5) {
            dim3 grid(1u << (log_h - 13), n_cols);
            LM_LAUNCH(ctx, k_ntt_second8, grid, dim3(512), 0, d_out, (const u32*)ctx->d_tw, (const u32*)ctx->d_tw_small, a);
        } else {
            dim3 grid(1u << (log_h - K - m), n_cols);
            size_t sh = ((1u << (K + m)) + (1u << K) + (K << m)) * 4;
            LM_LAUNCH(ctx, k_ntt_pass, grid, dim3(256), sh, d_in, d_out, ctx->d_tw, ctx->d_tw_small, a);
        }
        done += K;
        rem -= K;
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// =====================================================================================================
// Leaf sponge: digest(row) = hash_slice(row zero-padded to the full leaf width)
// (crates/backend/symetric/src/sponge.rs:7-24; crates/whir/src/merkle.rs:215-287).  One row per lane.
// =====================================================================================================
struct LeafArgs {
    u64 h;
    u32 is_ext;
    u32 stored_cols;   // EF: number of EF columns stored (plane stride); base: number of base columns stored
    u32 eff_words;     // words of the leaf that are backed by stored columns; the rest are zero
    u32 total_chunks;  // leaf_words / 8
    u32 data_chunks;   // chunks 0 .. data_chunks-1 may hold data
    u32 has_init;      // state after the all-zero suffix chunks was precomputed on the host
    u32 init[16];
};

__device__ __forceinline__ void load_chunk(const u32* __restrict__ mat, const LeafArgs& a, u64 row, u32 c, u32* dst) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 w = c * 8 + i;
        u32 v = 0;
        if (w < a.eff_words) {
            u32 colidx = a.is_ext ? (w % 5) * a.stored_cols + w / 5 : w;
            v = mat[(u64)colidx * a.h + row];
        }
        dst[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_leaf_sponge(const u32* __restrict__ mat, u32* __restrict__ digests, LeafArgs a) {
    u64 row = (u64)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.h) return;
    u32 s[16];
    int c;
    if (a.has_init) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = a.init[i];
        c = (int)a.data_chunks - 1;
    } else {
        load_chunk(mat, a, row, a.total_chunks - 2, s);
        load_chunk(mat, a, row, a.total_chunks - 1, s + 8);
        poseidon16_compress(s);
        c = (int)a.total_chunks - 3;
    }
    // software pipeline: the 8 words of the next chunk are requested before the current compression starts, so the
    // HBM latency of the column-strided loads hides under ~1.2 k modular multiplications
    u32 nxt[8];
    if (c >= 0) load_chunk(mat, a, row, (u32)c, nxt);
    for (; c >= 0; c--) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[8 + i] = nxt[i];
        if (c >= 1) load_chunk(mat, a, row, (u32)(c - 1), nxt);
        poseidon16_compress(s);
    }
    uint4* o = reinterpret_cast<uint4*>(digests + row * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// one Merkle level: next[i] = compress(prev[2i] || prev[2i+1])  (symetric/src/merkle.rs:50-90, compression.rs:5-15)
__global__ __launch_bounds__(256) void k_compress_layer(const u32* __restrict__ prev, u32* __restrict__ next, u64 n) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4* p = reinterpret_cast<const uint4*>(prev + i * 16);
    uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    u32 s[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    poseidon16_compress(s);
    uint4* o = reinterpret_cast<uint4*>(next + i * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

__global__ __launch_bounds__(256) void k_poseidon_batch(u32* __restrict__ states, u64 n, int compress) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4* p = reinterpret_cast<uint4*>(states + i * 16);
    uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    u32 s[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    if (compress)
        poseidon16_compress(s);
    else
        poseidon16_permute(s);
    p[0] = make_uint4(s[0], s[1], s[2], s[3]);
    p[1] = make_uint4(s[4], s[5], s[6], s[7]);
    p[2] = make_uint4(s[8], s[9], s[10], s[11]);
    p[3] = make_uint4(s[12], s[13], s[14], s[15]);
}

// ---- 16-lane cooperative variants (poseidon16_coop.h) for the latency-bound sizes ------------------------------------
// state i lives in lanes 16i .. 16i+15 of the grid
__global__ __launch_bounds__(256) void k_poseidon_batch_coop(u32* __restrict__ states, u64 n, int compress,
                                                             const u32* __restrict__ tab) {
    CoopRegs R;
    coop_load(R, tab);
    const u64 g = ((u64)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (g >= n) return;  // whole rows leave together (16 lanes share g)
    const u32 l = threadIdx.x & 15;
    const u32 x = states[g * 16 + l];
    states[g * 16 + l] = compress ? coop_compress(x, R) : coop_permute(x, R);
}
// 4-lane variant (poseidon16_quad.h): state i on lanes 4i .. 4i+3, four words each
__global__ __launch_bounds__(256) void k_poseidon_batch_quad(u32* __restrict__ states, u64 n, int compress, const u32* __restrict__ tab) {
    __shared__ u32 lds[QUAD_TAB_WORDS];
    quad_load_table(lds, tab);
    __syncthreads();
    const u64 g = ((u64)blockIdx.x * 256 + threadIdx.x) >> 2;
    if (g >= n) return;  // whole quads leave together
    const u32 q = threadIdx.x & 3;
    uint4* p = reinterpret_cast<uint4*>(states + g * 16 + 4 * q);
    const uint4 v = *p;
    u32 s[4] = {v.x, v.y, v.z, v.w};
    quad_permute(s, lds + q * QUAD_STRIDE);
    if (compress) s[0] = add(s[0], v.x), s[1] = add(s[1], v.y), s[2] = add(s[2], v.z), s[3] = add(s[3], v.w);
    *p = make_uint4(s[0], s[1], s[2], s[3]);
}
// one Merkle level, node i on lanes 16i .. 16i+15: the 16 input words are one coalesced 64-byte read
__global__ __launch_bounds__(256) void k_compress_layer_coop(const u32* __restrict__ prev, u32* __restrict__ next, u64 n,
                                                             const u32* __restrict__ tab) {
    CoopRegs R;
    coop_load(R, tab);
    const u64 g = ((u64)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (g >= n) return;
    const u32 l = threadIdx.x & 15;
    const u32 d = coop_compress(prev[g * 16 + l], R);
    if (l < 8) next[g * 8 + l] = d;
}
// up to five Merkle levels in one launch, node i on lanes 16i .. 16i+15: a workgroup owns 32 consecutive nodes of the input level
// and everything above them — 16, 8, 4, 2, 1 nodes — which it stores at their places in the (contiguous, bottom-up) digest layers and
// keeps in LDS for its next pass.  The levels of <= 2^14 nodes are latency: one launch per level was 6 us each whatever the size.
static constexpr u32 MID_SPAN = 32;
static bool merkle_mid_enabled() {  // LM_MERKLE_NO_MID=1: one launch per level (A/B measurements)
    static const bool on = getenv("LM_MERKLE_NO_MID") == nullptr;
    return on;
}
__global__ __launch_bounds__(256) void k_merkle_mid_coop(u32* __restrict__ lvl, u64 n_in, u32 levels, const u32* __restrict__ tab) {
    __shared__ u32 buf[2][MID_SPAN / 2 * 8];
    CoopRegs R;
    coop_load(R, tab);
    const u32 l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const u64 b = blockIdx.x;
    const u32* in = lvl + b * MID_SPAN * 8;
    u32* out = lvl + n_in * 8;  // the level being produced
    u64 n_out = n_in >> 1;
    for (u32 k = 0; k < levels; k++) {
        const u32 cnt = MID_SPAN >> (k + 1);  // nodes this workgroup produces at this level
        if (grp < cnt) {
            const u32 x = k == 0 ? in[grp * 16 + l] : buf[(k - 1) & 1][grp * 16 + l];
            const u32 d = coop_compress(x, R);
            if (l < 8) {
                out[(b * cnt + grp) * 8 + l] = d;
                buf[k & 1][grp * 8 + l] = d;
            }
        }
        __syncthreads();
        out += n_out * 8;
        n_out >>= 1;
    }
}
// the last levels of a tree in ONE workgroup of 256 lanes (16 nodes per pass): level sizes n0 >= n0/2 >= .. >= 1 nodes,
// level k read at lvl and written right behind it (digest layers are contiguous, bottom-up)
// The root is published through the pinned result buffer (sequence flag), so lm_commit needs no copy command.
__global__ __launch_bounds__(256) void k_merkle_top_coop(u32* __restrict__ lvl, u64 n_in, const u32* __restrict__ tab,
                                                         u32* __restrict__ h_res, u32 seq) {
    CoopRegs R;
    coop_load(R, tab);
    const u32 l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    u32* cur = lvl;
    for (u64 n = n_in >> 1; n >= 1; n >>= 1) {  // n = nodes of the level being produced
        u32* nxt = cur + 2 * n * 8;
        for (u64 g = grp; g < n; g += 16) {
            const u32 d = coop_compress(cur[g * 16 + l], R);
            if (l < 8) nxt[g * 8 + l] = d;
        }
        __threadfence_block();
        __syncthreads();
        cur = nxt;
    }
    if (threadIdx.x < 64) {
        if (threadIdx.x < 8) lm_store_system(h_res + threadIdx.x, cur[threadIdx.x]);
        lm_wait_stores();
        if (threadIdx.x == 0) lm_publish_flag(h_res, seq);
    }
}
// leaf sponge with one row per 16 lanes: lanes 0..7 carry the chaining value, lanes 8..15 fetch the next chunk
__global__ __launch_bounds__(256) void k_leaf_sponge_coop(const u32* __restrict__ mat, u32* __restrict__ digests, LeafArgs a,
                                                          const u32* __restrict__ tab) {
    CoopRegs R;
    coop_load(R, tab);
    const u64 row = ((u64)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (row >= a.h) return;
    const u32 l = threadIdx.x & 15;
    auto word = [&](u32 c, u32 i) -> u32 {  // word i of chunk c of this row (zero beyond the stored columns)
        const u32 w = c * 8 + i;
        if (w >= a.eff_words) return 0u;
        const u32 colidx = a.is_ext ? (w % 5) * a.stored_cols + w / 5 : w;
        return mat[(u64)colidx * a.h + row];
    };
    u32 s;
    int c;
    if (a.has_init) {
        s = a.init[l];
        c = (int)a.data_chunks - 1;
    } else {
        s = l < 8 ? word(a.total_chunks - 2, l) : word(a.total_chunks - 1, l - 8);
        s = coop_compress(s, R);
        c = (int)a.total_chunks - 3;
    }
    for (; c >= 0; c--) {
        if (l >= 8) s = word((u32)c, l - 8);
        s = coop_compress(s, R);
    }
    if (l < 8) digests[row * 8 + l] = s;
}

// Poseidon16 table rows (reference: generate_trace_rows_for_perm, crates/lean_vm/src/tables/poseidon_16/trace_gen.rs:44-112).
// One row per lane: reads the 16 input columns and flag_permute, writes the 84 derived columns
// (2 x 16 post-states of the initial full-round pairs, 20 partial-round S-box outputs, 16 post-state of the first terminal
// pair, 8 + 8 outputs).  cols: device array of 109 column pointers (Poseidon1Cols16 order, poseidon_16/mod.rs:366-383).
__global__ __launch_bounds__(256) void k_poseidon_trace(u32* const* __restrict__ cols, u64 n_rows) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const PoseidonConsts& K = poseidon_consts();
    u32 in[16], s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) in[i] = s[i] = cols[9 + i][r];
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], K.rc_init[2 * blk + h][i]));
            mds_circ16(s);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) cols[25 + 16 * blk + i][r] = s[i];
    }
    {
        u32 t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) t[i] = s[i];
#pragma unroll 1
        for (int i = 0; i < 16; i++) s[i] = add(dot16(t, K.dmat[i]), K.dbias[i]);
    }
#pragma unroll 1
    for (int pr = 0; pr < 20; pr++) {
        u32 s0 = cube(s[0]);
        cols[57 + pr][r] = s0;
        if (pr < 19) s0 = add(s0, K.pscalar[pr]);
        s[0] = s0;
        const u32 n0 = dot16(s, K.prow[pr]);
#pragma unroll
        for (int i = 1; i < 16; i++) s[i] = add(s[i], mul(s0, K.pcol[pr][i - 1]));
        s[0] = n0;
    }
#pragma unroll 1
    for (int h = 0; h < 4; h++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], K.rc_term[h][i]));
        mds_circ16(s);
        if (h == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) cols[77 + i][r] = s[i];
        }
    }
    const u32 fp = cols[8][r];  // flag_permute
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 cv = add(s[i], in[i]);
        cols[93 + i][r] = add(mul(sub(ONE, fp), cv), mul(fp, s[i]));
        cols[101 + i][r] = mul(fp, s[i + 8]);
    }
}

// batched opening: block b serves indices[b].  PINNED: leaves / siblings are host memory (the staging ring): written with
// system-scope stores, and the block that finishes last publishes the sequence number — no copy commands, no synchronise.
template <bool PINNED>
__global__ __launch_bounds__(256) void k_tree_open(const u32* __restrict__ mat, const u32* __restrict__ digests,
                                                   const u64* __restrict__ indices, u32* __restrict__ leaves,
                                                   u32* __restrict__ siblings, u64 h, u32 log_h, u32 is_ext,
                                                   u32 stored_cols, u32 eff_words, u32 leaf_words, u32* __restrict__ done_counter,
                                                   u32* __restrict__ h_res, u32 seq) {
    const u64 idx = indices[blockIdx.x];
    for (u32 w = threadIdx.x; w < leaf_words; w += 256) {
        u32 v = 0;
        if (w < eff_words) {
            u32 colidx = is_ext ? (w % 5) * stored_cols + w / 5 : w;
            v = mat[(u64)colidx * h + idx];
        }
        if (PINNED)
            lm_store_system(leaves + (u64)blockIdx.x * leaf_words + w, v);
        else
            leaves[(u64)blockIdx.x * leaf_words + w] = v;
    }
    for (u32 x = threadIdx.x; x < log_h * 8; x += 256) {
        u32 lvl = x >> 3, k = x & 7;
        // layer lvl starts at offset sum_{i<lvl} (h >> i) = 2h - (h >> (lvl-1)) ... computed incrementally
        u64 off = 2 * h - (2 * h >> lvl);
        u64 node = (idx >> lvl) ^ 1;
        const u32 v = digests[(off + node) * 8 + k];
        if (PINNED)
            lm_store_system(siblings + (u64)blockIdx.x * log_h * 8 + x, v);
        else
            siblings[(u64)blockIdx.x * log_h * 8 + x] = v;
    }
    if (PINNED) {
        lm_wait_stores();
        __syncthreads();
        if (threadIdx.x == 0 && lm_ticket(done_counter) == gridDim.x - 1) {
            lm_store_agent(done_counter, 0);
            lm_publish_flag(h_res, seq);
        }
    }
}

extern "C" {

// Below this many independent permutations the chip is not full with one permutation per lane and a launch lasts one
// ~17 us dependent chain regardless of n: the 16-lane variant (3 us chain, 2.5x the instructions) is faster.
static constexpr u64 COOP_MAX_PERMS = 1ull << 14;

static int poseidon_batch(lm_ctx* ctx, uint32_t* d_states, uint64_t n, int compress) {
    LM_REQUIRE(ctx && d_states);
    if (n == 0) return LM_OK;
    if (n <= COOP_MAX_PERMS)
        LM_LAUNCH(ctx, k_poseidon_batch_coop, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, d_states, n, compress,
                  (const u32*)ctx->d_coop);
    else
        LM_LAUNCH(ctx, k_poseidon_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_states, n, compress);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int lm_poseidon16_permute(lm_ctx* ctx, uint32_t* d_states, uint64_t n) { return poseidon_batch(ctx, d_states, n, 0); }
int lm_poseidon16_permute_quad(lm_ctx* ctx, uint32_t* d_states, uint64_t n, int compress) {
    LM_REQUIRE(ctx && d_states);
    if (n == 0) return LM_OK;
    LM_LAUNCH(ctx, k_poseidon_batch_quad, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, d_states, n, compress, (const u32*)ctx->d_quad);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int lm_poseidon16_compress(lm_ctx* ctx, uint32_t* d_states, uint64_t n) { return poseidon_batch(ctx, d_states, n, 1); }

int lm_poseidon_trace(lm_ctx* ctx, uint32_t* const* d_cols, uint64_t n_rows) {
    LM_REQUIRE(ctx && d_cols);
    if (n_rows == 0) return LM_OK;
    u32* s;
    int rc = lm_scratch(ctx, 109 * 2 + 16, &s);
    if (rc) return rc;
    if ((rc = lm_stage_upload(ctx, s, d_cols, 109 * sizeof(u32*)))) return rc;
    LM_LAUNCH(ctx, k_poseidon_trace, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (u32* const*)s, n_rows);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

int lm_commit(lm_ctx* ctx, const uint32_t* d_evals, int is_ext, uint32_t n_vars, uint32_t folding_factor,
              uint32_t log_inv_rate, uint64_t actual_len, lm_tree** out, uint32_t root[LM_DIGEST_WORDS]) {
    LM_REQUIRE(ctx && d_evals && out && root);
    LM_REQUIRE(folding_factor >= 1 && folding_factor <= n_vars);
    LM_REQUIRE(n_vars + log_inv_rate - folding_factor <= LM_TW_LOG);
    const u64 len = 1ull << n_vars;
    LM_REQUIRE(actual_len >= 1 && actual_len <= len);
    const u32 n_cols = 1u << folding_factor;
    const u32 dim = is_ext ? 5 : 1;
    LM_REQUIRE((n_cols * dim) % 8 == 0 && n_cols * dim >= 16);
    const u64 col_len = len >> folding_factor;
    const u32 log_h = n_vars + log_inv_rate - folding_factor;
    const u64 h = 1ull << log_h;
    // WhirConfig::commit, commit.rs:70-73
    const u32 eff_cols = (u32)((actual_len + col_len - 1) / col_len);

    lm_tree* t = new lm_tree();
    t->log_h = log_h;
    t->is_ext = is_ext ? 1 : 0;
    t->n_cols = n_cols;
    t->eff_cols = eff_cols;
    t->stored_words = eff_cols * dim;
    t->leaf_words = n_cols * dim;
    if (lm_pool_alloc_t(ctx, &t->d_matrix, (u64)t->stored_words * h * 4) != hipSuccess ||
        lm_pool_alloc_t(ctx, &t->d_digests, (2 * h - 1) * 8 * 4) != hipSuccess) {
        lm_set_error("lm_commit: hipMalloc failed (%llu + %llu bytes)", (unsigned long long)t->stored_words * h * 4,
                     (unsigned long long)(2 * h - 1) * 32);
        lm_pool_free(ctx, t->d_matrix);
        delete t;
        return LM_E_NOMEM;
    }
    int rc;
    // EF: SoA planes, plane k holds 2^n_vars words; its first eff_cols column slices are stored at k*eff_cols + c
    rc = lde_columns(ctx, d_evals, t->d_matrix, eff_cols, is_ext ? 5 : 1, len, log_h, log_inv_rate);
    if (rc != LM_OK) {
        lm_tree_free(ctx, t);
        return rc;
    }
    // leaf digests
    LeafArgs la;
    memset(&la, 0, sizeof la);
    la.h = h;
    la.is_ext = t->is_ext;
    la.stored_cols = eff_cols;
    la.eff_words = t->stored_words;
    la.total_chunks = t->leaf_words / 8;
    const u32 zero_chunks = (t->leaf_words - t->stored_words) / 8;  // merkle.rs:65
    if (zero_chunks >= 2) {
        // precompute_zero_suffix_state, sponge.rs:27-49 (host Poseidon, row independent)
        u32 st[16];
        memset(st, 0, sizeof st);
        poseidon16_compress(st);
        for (u32 i = 0; i + 2 < zero_chunks; i++) {
            for (int j = 8; j < 16; j++) st[j] = 0;
            poseidon16_compress(st);
        }
        la.has_init = 1;
        memcpy(la.init, st, sizeof st);
        la.data_chunks = la.total_chunks - zero_chunks;
    } else {
        la.has_init = 0;
        la.data_chunks = la.total_chunks;
    }
    const u32* coop = ctx->d_coop;
    if (h <= COOP_MAX_PERMS)
        LM_LAUNCH(ctx, k_leaf_sponge_coop, dim3((unsigned)((h * 16 + 255) / 256)), dim3(256), 0, (const u32*)t->d_matrix,
                  t->d_digests, la, coop);
    else {
        LM_LAUNCH(ctx, k_leaf_sponge, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, t->d_matrix, t->d_digests, la);
        // (profile leg of bench.py: the "bytes" of this kernel are its PERMUTATIONS — the first two chunks of a row share one)
        LM_PROF_BYTES(ctx, k_leaf_sponge, (u64)h * (la.has_init ? la.data_chunks : la.total_chunks - 1));
    }
    // levels: one permutation per lane while the level fills the chip, 16 lanes per node below that, and the last levels
    // (<= 64 nodes) in a single workgroup
    u64 off = 0;
    u32 root_seq = 0;
    for (u64 n = h; n > 1; n >>= 1) {
        u64 next_n = n >> 1;
        if (next_n <= 64) {
            root_seq = ++ctx->res_seq;
            LM_LAUNCH(ctx, k_merkle_top_coop, dim3(1), dim3(256), 0, t->d_digests + off * 8, n, coop, ctx->h_res, root_seq);
            break;
        }
        if (next_n <= COOP_MAX_PERMS && n >= 2 * MID_SPAN && merkle_mid_enabled()) {
            // several levels per launch while the last one keeps >= 128 nodes for the single-workgroup top
            u32 levels = 0;
            while (levels < 4 && (n >> (levels + 1)) >= 128) levels++;
            if (levels) {
                LM_LAUNCH(ctx, k_merkle_mid_coop, dim3((unsigned)(n / MID_SPAN)), dim3(256), 0, t->d_digests + off * 8, n, levels, coop);
                for (u32 k = 0; k + 1 < levels; k++) off += n, n >>= 1;  // (the loop's own step accounts for the last level)
                off += n;
                continue;
            }
        }
        if (next_n <= COOP_MAX_PERMS)
            LM_LAUNCH(ctx, k_compress_layer_coop, dim3((unsigned)((next_n * 16 + 255) / 256)), dim3(256), 0,
                      (const u32*)(t->d_digests + off * 8), t->d_digests + (off + n) * 8, next_n, coop);
        else
            LM_LAUNCH(ctx, k_compress_layer, dim3((unsigned)((next_n + 255) / 256)), dim3(256), 0,
                      t->d_digests + off * 8, t->d_digests + (off + n) * 8, next_n);
        off += n;
    }
    bool ok = hipGetLastError() == hipSuccess;
    if (ok && root_seq) {
        ok = lm_wait_result(ctx, root_seq) == LM_OK;
        if (ok) memcpy(root, ctx->h_res, 32);
    } else if (ok) {  // single-row tree: the leaf digest is the root
        ok = hipMemcpyAsync(root, t->d_digests + (2 * h - 2) * 8, 32, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) {
        lm_set_error("lm_commit: kernel launch / sync failed: %s", hipGetErrorString(hipGetLastError()));
        lm_tree_free(ctx, t);
        return LM_E_DEVICE;
    }
    *out = t;
    return LM_OK;
}

void lm_tree_free(lm_ctx* ctx, lm_tree* t) {
    if (!t) return;
    lm_pool_free(ctx, t->d_matrix);
    lm_pool_free(ctx, t->d_digests);
    delete t;
}
uint32_t lm_tree_log_height(const lm_tree* t) { return t ? t->log_h : 0; }
uint32_t lm_tree_leaf_words(const lm_tree* t) { return t ? t->leaf_words : 0; }

// The opening as two calls: the kernel is enqueued with its results going into pinned host memory behind a published sequence
// number, the caller does other work — WHIR enqueues the weight kernels of the same round, which depend on the query INDICES only —
// and collects leaves and paths afterwards.  Work enqueued on the context's stream in between is ordered behind the opening kernel
// (the scratch buffer both use is reused in stream order).  Without room in the staging ring the opening completes inside _begin.
struct lm_tree_opening {
    u32 seq = 0;
    u64 leaf_total = 0, sib_total = 0;
    const u32 *p_leaves = nullptr, *p_sib = nullptr;  // pinned (seq != 0)
    std::vector<u32> done;                            // leaves | siblings, already on the host (seq == 0)
};
int lm_tree_open_begin(lm_ctx* ctx, const lm_tree* t, const uint64_t* indices, uint32_t n_idx, lm_tree_opening** out) {
    LM_REQUIRE(ctx && t && indices && out && n_idx >= 1);
    const u64 h = 1ull << t->log_h;
    for (u32 i = 0; i < n_idx; i++) LM_REQUIRE(indices[i] < h);
    const u64 leaf_total = (u64)n_idx * t->leaf_words, sib_total = (u64)n_idx * t->log_h * 8;
    u32* d_tmp;
    int rc = lm_scratch(ctx, 2ull * n_idx + leaf_total + sib_total, &d_tmp);
    if (rc) return rc;
    u64* d_idx = reinterpret_cast<u64*>(d_tmp);
    u32* d_leaves = d_tmp + 2ull * n_idx;
    u32* d_sib = d_leaves + leaf_total;
    if ((rc = lm_stage_upload(ctx, d_idx, indices, (size_t)n_idx * 8))) return rc;
    void* pinned;
    if ((rc = lm_stage_alloc(ctx, (leaf_total + sib_total) * 4, &pinned))) return rc;
    std::unique_ptr<lm_tree_opening> o(new lm_tree_opening());
    o->leaf_total = leaf_total, o->sib_total = sib_total;
    if (pinned) {  // results straight into pinned host memory
        u32* p_leaves = static_cast<u32*>(pinned);
        u32* p_sib = p_leaves + leaf_total;
        o->seq = ++ctx->res_seq;
        o->p_leaves = p_leaves, o->p_sib = p_sib;
        LM_LAUNCH(ctx, k_tree_open<true>, dim3(n_idx), dim3(256), 0, t->d_matrix, t->d_digests, d_idx, p_leaves, p_sib, h, t->log_h,
                  t->is_ext, t->eff_cols, t->stored_words, t->leaf_words, ctx->d_sync + 1, ctx->h_res, o->seq);
        LM_HIP(hipGetLastError());
    } else {
        LM_LAUNCH(ctx, k_tree_open<false>, dim3(n_idx), dim3(256), 0, t->d_matrix, t->d_digests, d_idx, d_leaves, d_sib, h, t->log_h,
                  t->is_ext, t->eff_cols, t->stored_words, t->leaf_words, (u32*)nullptr, (u32*)nullptr, 0u);
        LM_HIP(hipGetLastError());
        o->done.resize(leaf_total + sib_total);
        LM_HIP(hipMemcpyAsync(o->done.data(), d_leaves, (leaf_total + sib_total) * 4, hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = o.release();
    return LM_OK;
}
// consumes the handle (also on failure; leaves = NULL abandons the opening)
int lm_tree_open_end(lm_ctx* ctx, lm_tree_opening* opening, uint32_t* leaves, uint32_t* siblings) {
    if (!opening) return LM_OK;
    std::unique_ptr<lm_tree_opening> o(opening);
    LM_REQUIRE(ctx);
    if (o->seq) {  // (an abandoned opening still writes into the staging ring: let it finish)
        int rc = lm_wait_result(ctx, o->seq);
        if (rc) return rc;
    }
    if (!leaves || !siblings) return LM_OK;
    const u32* src_l = o->seq ? o->p_leaves : o->done.data();
    const u32* src_s = o->seq ? o->p_sib : o->done.data() + o->leaf_total;
    memcpy(leaves, src_l, o->leaf_total * 4);
    if (o->sib_total) memcpy(siblings, src_s, o->sib_total * 4);
    return LM_OK;
}
int lm_tree_open(lm_ctx* ctx, const lm_tree* t, const uint64_t* indices, uint32_t n_idx, uint32_t* leaves,
                 uint32_t* siblings) {
    LM_REQUIRE(ctx && t && indices && leaves && siblings);
    if (n_idx == 0) return LM_OK;
    lm_tree_opening* o = nullptr;
    int rc = lm_tree_open_begin(ctx, t, indices, n_idx, &o);
    if (rc) return rc;
    return lm_tree_open_end(ctx, o, leaves, siblings);
}

int lm_tree_download_matrix(lm_ctx* ctx, const lm_tree* t, uint32_t* rows) {
    LM_REQUIRE(ctx && t && rows);
    const u64 h = 1ull << t->log_h;
    std::vector<u32> cm((u64)t->stored_words * h);
    LM_HIP(hipMemcpyAsync(cm.data(), t->d_matrix, cm.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    for (u64 r = 0; r < h; r++)
        for (u32 w = 0; w < t->leaf_words; w++) {
            u32 v = 0;
            if (w < t->stored_words) {
                u32 colidx = t->is_ext ? (w % 5) * t->eff_cols + w / 5 : w;
                v = cm[(u64)colidx * h + r];
            }
            rows[r * t->leaf_words + w] = v;
        }
    return LM_OK;
}
int lm_tree_download_digests(lm_ctx* ctx, const lm_tree* t, uint32_t* digests) {
    LM_REQUIRE(ctx && t && digests);
    const u64 h = 1ull << t->log_h;
    LM_HIP(hipMemcpyAsync(digests, t->d_digests, (2 * h - 1) * 32, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    return LM_OK;
}

}  // extern "C"
