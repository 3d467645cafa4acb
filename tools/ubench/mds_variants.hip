// micro-benchmark: circulant MDS variants (G mds/s chip-wide) — correctness checked against the product's mds_circ16
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using kb::u32; using kb::u64;
__constant__ u32 kC[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};

// V1: every product through v_mad_u64_u32 (constants from scalar memory, no strength reduction)
__device__ __forceinline__ void mds_v1(u32 s[16]) {
    u32 o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) acc += (u64)s[j] * kC[(16 + i - j) & 15];
        o[i] = kb::reduce40(acc);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = o[i];
}
// V2: 16-bit halves, 24-bit multiplies (full rate), 32-bit accumulators
__device__ __forceinline__ u32 reduce_halves(u32 sh, u32 sl) {
    // value = sh * 2^16 + sl,  sh < 2^24, sl < 2^25.   2^31 = 2^24 - 1 (mod p)
    // sh = a * 2^15 + b  ->  sh * 2^16 = a * 2^31 + b * 2^16 = a * (2^24 - 1) + b * 2^16
    u32 a = sh >> 15, b = sh & 0x7fffu;                 // a < 2^9
    u32 a1 = a >> 7, a0 = a & 127u;                      // a * 2^24 = a1 * 2^31 + a0 * 2^24 = a1 (2^24 - 1) + a0 * 2^24
    u32 t1 = (a0 << 24) + sl + a1 * 0x00ffffffu;         // < 2^31 + 2^25 + 2^26
    u32 t2 = (b << 16) + (kb::P - a);                    // < 2^31 + 2^31
    t1 = kb::umin(t1, t1 - kb::P);                       // < 2^31 (t1 < 2p)
    t2 = kb::umin(t2, t2 - kb::P);
    t1 = kb::umin(t1, t1 - kb::P);
    t2 = kb::umin(t2, t2 - kb::P);
    return kb::add(t1, t2);
}
__device__ __forceinline__ void mds_v2(u32 s[16]) {
    constexpr u32 C[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    u32 lo[16], hi[16], o[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { lo[j] = s[j] & 0xffffu; hi[j] = s[j] >> 16; }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        u32 sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u32 c = C[(16 + i - j) & 15];
            sl = __umul24(lo[j], c) + sl;
            sh = __umul24(hi[j], c) + sh;
        }
        o[i] = reduce_halves(sh, sl);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = o[i];
}
template <int V>
__global__ __launch_bounds__(256) void k_mds(u32* out, u32 seed, int reps) {
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = (seed * 2654435761u + threadIdx.x * 16 + i + blockIdx.x * 7919u) % kb::P;
    for (int r = 0; r < reps; r++) {
        if (V == 0) kb::mds_circ16(s);
        if (V == 1) mds_v1(s);
        if (V == 2) mds_v2(s);
    }
    u32 x = 0;
    for (int i = 0; i < 16; i++) x = x * 31 + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
int main() {
    const int blocks = 8192;
    u32* d; hipMalloc(&d, blocks * 256 * 4 * 3);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static u32 h[3][8192 * 256];
    for (int v = 0; v < 3; v++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_mds<0>, dim3(blocks), dim3(256), 0, 0, d + v * blocks * 256, 7u, 64);
            if (v == 1) hipLaunchKernelGGL(k_mds<1>, dim3(blocks), dim3(256), 0, 0, d + v * blocks * 256, 7u, 64);
            if (v == 2) hipLaunchKernelGGL(k_mds<2>, dim3(blocks), dim3(256), 0, 0, d + v * blocks * 256, 7u, 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h[v], d + v * blocks * 256, blocks * 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < blocks * 256; i++) bad += h[v][i] != h[0][i];
        printf("variant %d: %.3f ms  %.1f G mds/s  mismatches vs V0: %d\n", v, ms, (double)blocks * 256 * 64 / ms / 1e6, bad);
    }
}
