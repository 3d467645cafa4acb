// micro-benchmark (round-3 review, item 4): the 16 x 16 circulant MDS of Poseidon1-16 as an i8 MFMA on byte planes.
//   out_i = sum_j C[(i - j) & 15] s_j with C < 2^7 and s_j < p < 2^31.  With s_j = sum_k 256^k d_k[j] in SIGNED digits
//   (t = s + 0x00808080, d_k = byte_k(t) - 128 for k < 3, d_3 = byte_3(t) <= 127: bytes of t ^ 0x00808080 read as i8), the product is
//   four 16 x 16 by 16 x N integer products M d_k — v_mfma_i32_16x16x32_i8 with the upper half of K zero (one plane of 16 permutations
//   per instruction) — recombined as sum_k 256^k (M d_k) < 2^40 and reduced.
// The sponge keeps one permutation per LANE (16 state words in registers); the MFMA wants a column of 8 consecutive state bytes per
// lane and returns 4 output rows per lane, so the state crosses LDS on the way in (digits: a 4 x 4 byte transpose per 4 words, then
// lane (g, n) of batch q reads the bytes of words 8g .. 8g+7 of lane 16q + n) and on the way out.  Variants:
//   0  mds_circ16 of poseidon16.h (the production form: one CRT split, 128 multiply-adds + 16 reductions per permutation)
//   1  MFMA with the conversions (what the sponge kernel would have to do)
//   2  MFMA alone on operands that are already in its layout, recombination and reduction included (upper bound)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mds_mfma.hip -o mds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using kb::u32;
using kb::u64;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int LSTRIDE = 18;  // words per lane in the LDS staging area (16 + 2: the 8-byte reads of 16 lanes spread over the banks)

__device__ __forceinline__ constexpr u32 ccol(int m) {
    constexpr u32 C[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    return C[m & 15];
}
// A operand of lane (g, m): M[m][8g .. 8g+7] = C[(m - k) & 15] for g < 2, zero above (K = 32, 16 used)
__device__ __forceinline__ long a_operand(u32 lane) {
    const u32 m = lane & 15, g = lane >> 4;
    u64 a = 0;
    if (g < 2)
        for (int b = 0; b < 8; b++) a |= (u64)ccol((int)m - (int)(8 * g + b) + 32) << (8 * b);
    return (long)a;
}
// digits of 4 words as 4 words of bytes: X[p] = (d_p[w0], d_p[w1], d_p[w2], d_p[w3])
__device__ __forceinline__ void digit_planes(u32 a, u32 b, u32 c, u32 d, u32 (&X)[4]) {
    a = (a + 0x00808080u) ^ 0x00808080u, b = (b + 0x00808080u) ^ 0x00808080u, c = (c + 0x00808080u) ^ 0x00808080u, d = (d + 0x00808080u) ^ 0x00808080u;
    const u32 ab02 = __builtin_amdgcn_perm(b, a, 0x06020400u), ab13 = __builtin_amdgcn_perm(b, a, 0x07030501u);
    const u32 cd02 = __builtin_amdgcn_perm(d, c, 0x06020400u), cd13 = __builtin_amdgcn_perm(d, c, 0x07030501u);
    X[0] = __builtin_amdgcn_perm(cd02, ab02, 0x05040100u);
    X[2] = __builtin_amdgcn_perm(cd02, ab02, 0x07060302u);
    X[1] = __builtin_amdgcn_perm(cd13, ab13, 0x05040100u);
    X[3] = __builtin_amdgcn_perm(cd13, ab13, 0x07060302u);
}
// sum_k 256^k y_k, y_k exact integers (the total is M s >= 0, < 2^40), then mod p
__device__ __forceinline__ u32 recombine(int y0, int y1, int y2, int y3) {
    const long long t = (long long)y0 + ((long long)y1 << 8) + ((long long)y2 << 16) + ((long long)y3 << 24);
    return kb::reduce40((u64)t);
}

// variant 1: state in registers (one permutation per lane) -> MFMA -> state in registers
__device__ __forceinline__ void mds_mfma(u32 (&s)[16], u32* lds /* this wave's 64 * LSTRIDE words */, long a_op, u32 lane) {
    const u32 g = lane >> 4, n = lane & 15;
    // digits, transposed: word (p, grp) of a lane = digit p of its state words 4 grp .. 4 grp + 3
#pragma unroll
    for (int grp = 0; grp < 4; grp++) {
        u32 X[4];
        digit_planes(s[4 * grp], s[4 * grp + 1], s[4 * grp + 2], s[4 * grp + 3], X);
#pragma unroll
        for (int p = 0; p < 4; p++) lds[lane * LSTRIDE + p * 4 + grp] = X[p];
    }
    __builtin_amdgcn_wave_barrier();
    int y[4][4][4];  // [batch q][plane p][row i]
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint2 bw = *reinterpret_cast<const uint2*>(lds + (16 * q + n) * LSTRIDE + p * 4 + 2 * (g & 1));
            const long b_op = g < 2 ? (long)(((u64)bw.y << 32) | bw.x) : 0l;
            const v4i acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_op, b_op, v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) y[q][p][i] = acc[i];
        }
    __builtin_amdgcn_wave_barrier();
    // lane (g, n) holds rows 4g .. 4g+3 of permutation 16q + n: back through LDS to one permutation per lane
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) lds[(16 * q + n) * LSTRIDE + 4 * g + i] = recombine(y[q][0][i], y[q][1][i], y[q][2][i], y[q][3][i]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] = lds[lane * LSTRIDE + j];
    __builtin_amdgcn_wave_barrier();
}

template <int V>
__global__ __launch_bounds__(256) void k_mds(u32* out, u32 seed, int iters) {
    __shared__ u32 lds_all[4 * 64 * LSTRIDE];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32* lds = lds_all + wave * 64 * LSTRIDE;
    const long a_op = a_operand(lane);
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = (u32)(((u64)(seed + i) * 2654435761u + (blockIdx.x * 256 + threadIdx.x) * 40503u) % kb::P);
    if (V == 2) {
        // operands in the MFMA's own layout: this lane's B words of the 4 planes stay in registers, the outputs feed the next
        // iteration's digits as they are (not the same function as variants 0 / 1: a rate, not a value)
        u32 w[4] = {s[0], s[1], s[2], s[3]};
        for (int it = 0; it < iters; it++) {
            u32 X[4];
            digit_planes(w[0], w[1], w[2], w[3], X);
            int y[4][4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const long b_op = (long)(((u64)X[(p + 1) & 3] << 32) | X[p]);
                const v4i acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_op, b_op, v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) y[p][i] = acc[i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = recombine(y[0][i] & 0xffffff, y[1][i] & 0xffffff, y[2][i] & 0xffffff, y[3][i] & 0x7fff);
        }
        out[blockIdx.x * 256 + threadIdx.x] = w[0] ^ w[1] ^ w[2] ^ w[3];
        return;
    }
    for (int it = 0; it < iters; it++) {
        if (V == 0) kb::mds_circ16(s);
        if (V == 1) mds_mfma(s, lds, a_op, lane);
    }
    u32 x = 0;
    for (int i = 0; i < 16; i++) x = x * 31 + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
    const int blocks = 8192, iters = 64;
    u32* d;
    if (hipMalloc(&d, (size_t)blocks * 256 * 4 * 3) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    static u32 h[2][8192 * 256];
    for (int v = 0; v < 3; v++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_mds<0>, dim3(blocks), dim3(256), 0, 0, d + (size_t)v * blocks * 256, 7u, iters);
            if (v == 1) hipLaunchKernelGGL(k_mds<1>, dim3(blocks), dim3(256), 0, 0, d + (size_t)v * blocks * 256, 7u, iters);
            if (v == 2) hipLaunchKernelGGL(k_mds<2>, dim3(blocks), dim3(256), 0, 0, d + (size_t)v * blocks * 256, 7u, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        if (hipGetLastError() != hipSuccess) printf("launch error\n");
        int bad = -1;
        if (v < 2) {
            hipMemcpy(h[v], d + (size_t)v * blocks * 256, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost);
            bad = 0;
            for (int i = 0; i < blocks * 256; i++) bad += h[v][i] != h[0][i];
        }
        // variant 2 processes 16 permutations' MDS per 4 MFMA per wave iteration: a wave iteration = 16 MDS (not 64)
        const double mds = v == 2 ? (double)blocks * 4 * 16 * iters : (double)blocks * 256 * iters;
        printf("variant %d (%s): %.3f ms  %.1f G MDS/s", v, v == 0 ? "mds_circ16, VALU" : v == 1 ? "i8 MFMA with layout conversions" : "i8 MFMA, operands in place", ms,
               mds / ms / 1e6);
        if (bad >= 0) printf("  mismatches vs variant 0: %d", bad);
        printf("\n");
    }
    return 0;
}
