// micro-benchmark: is the circulant MDS cheaper through byte planes and v_dot4_u32_u8 than through 256 v_mad_u64_u32?
//   out_i = sum_j C[(i-j)&15] s_j,  s_j = sum_p 2^(8p) b_p[j]  =>  out_i = sum_p 2^(8p) sum_g dot4(X[p][g], K[(i-4g)&15])
// with X[p][g] = bytes p of s[4g..4g+3] (a 4x4 byte transpose per group, v_perm_b32) and K[m] = (C[m], C[m-1], C[m-2], C[m-3]).
// Prints rates of v_dot4_u32_u8 / v_perm_b32 and of both MDS forms, and checks that they agree.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mds_dot4.hip -o mds_dot4
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using kb::u32; using kb::u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k_rate(u32* out, u32 seed) {
    u32 a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
    u32 c = seed | 0x01010101;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_amdgcn_udot4(a[i], c, a[i], false);   // v_dot4_u32_u8
            if (OP == 1) a[i] = __builtin_amdgcn_perm(a[i], c, 0x05010400u + i); // v_perm_b32
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ constexpr u32 kpack(int m) {
    constexpr u32 C[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    return C[m & 15] | (C[(m + 15) & 15] << 8) | (C[(m + 14) & 15] << 16) | (C[(m + 13) & 15] << 24);
}
__device__ __forceinline__ void mds_dot4(u32 s[16]) {
    // byte transposes: X[p][g] = byte p of s[4g], s[4g+1], s[4g+2], s[4g+3]
    u32 X[4][4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const u32 a = s[4 * g], b = s[4 * g + 1], c = s[4 * g + 2], d = s[4 * g + 3];
        // v_perm_b32(hi, lo, sel): byte k of the result = byte sel_k of (hi:lo) (lo = bytes 0..3, hi = bytes 4..7)
        const u32 ab02 = __builtin_amdgcn_perm(b, a, 0x06020400u);  // a0 b0 a2 b2   (bytes: a0, b0(4), a2, b2(6))
        const u32 ab13 = __builtin_amdgcn_perm(b, a, 0x07030501u);  // a1 b1 a3 b3
        const u32 cd02 = __builtin_amdgcn_perm(d, c, 0x06020400u);
        const u32 cd13 = __builtin_amdgcn_perm(d, c, 0x07030501u);
        X[0][g] = __builtin_amdgcn_perm(cd02, ab02, 0x05040100u);  // a0 b0 c0 d0
        X[2][g] = __builtin_amdgcn_perm(cd02, ab02, 0x07060302u);  // a2 b2 c2 d2
        X[1][g] = __builtin_amdgcn_perm(cd13, ab13, 0x05040100u);
        X[3][g] = __builtin_amdgcn_perm(cd13, ab13, 0x07060302u);
    }
    u32 o[16];
    kb::static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        u32 acc[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            u32 t = 0;
            kb::static_for<0, 4>([&](auto G) {
                constexpr int g = decltype(G)::value;
                u32 k = kpack(i - 4 * g + 64);
                asm("" : "+s"(k));
                t = __builtin_amdgcn_udot4(X[p][g], k, t, false);
            });
            acc[p] = t;
        }
        const u32 L = acc[0] + (acc[1] << 8), H = acc[2] + (acc[3] << 8);
        o[i] = kb::reduce40((u64)H * kb::opaque_const(65536u) + L);
    });
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = o[i];
}
template <int V>
__global__ __launch_bounds__(256) void k_mds(u32* out, u32 seed, int reps) {
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = ((seed + threadIdx.x * 16 + i) * 2654435761u) % kb::P;
    for (int r = 0; r < reps; r++) {
        if (V == 0) kb::mds_circ16(s); else mds_dot4(s);
    }
    u32 x = 0;
    for (int i = 0; i < 16; i++) x = x * 31 + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
int main() {
    const int blocks = 8192;
    u32 *d, *d2; CHECK(hipMalloc(&d, blocks * 256 * 4)); CHECK(hipMalloc(&d2, blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    for (int op = 0; op < 2; op++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s %8.3f ms  %7.2f T lane-ops/s\n", op == 0 ? "v_dot4_u32_u8" : "v_perm_b32", ms, (double)blocks * 256 * 8 * ITERS / ms / 1e9);
    }
    for (int v = 0; v < 2; v++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_mds<0>, dim3(blocks), dim3(256), 0, 0, d, 7u, 64);
            else hipLaunchKernelGGL(k_mds<1>, dim3(blocks), dim3(256), 0, 0, d2, 7u, 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s %8.3f ms  %7.2f G mds/s\n", v == 0 ? "mds_circ16 (mad64)" : "mds via dot4", ms, (double)blocks * 256 * 64 / ms / 1e6);
    }
    static u32 h[2][4096];
    CHECK(hipMemcpy(h[0], d, sizeof h[0], hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h[1], d2, sizeof h[1], hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 4096; i++) bad += h[0][i] != h[1][i];
    printf("mismatching lanes (of 4096): %d\n", bad);
    return 0;
}
