// micro-benchmark: Poseidon1-16 permutation variants (G perm/s chip-wide), checked against the product's permutation
//   V0 product poseidon16_permute
//   V1 partial block through affine forms, tables in __constant__ memory (scalar loads)
//   V2 same, tables constexpr (instruction literals)
//   V3 same, tables staged in LDS (broadcast ds_read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using namespace kb;
struct Lin { u32 y[20][37]; u32 fin[16][37]; };
__constant__ Lin kLinC =
#include "../../leanmultisig_amd/csrc/poseidon16_linear_hash.inc"
;
static constexpr Lin kLinL =
#include "../../leanmultisig_amd/csrc/poseidon16_linear_hash.inc"
;
template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (I < N) { f(IC<I>{}); sfor<I + 1, N>(f); } }

template <int MODE>
__device__ __forceinline__ void perm_lin(u32 s[16], const Lin* lds) {
    const PoseidonConsts& K = poseidon_consts();
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], K.rc_init[r][i]));
        mds_circ16(s);
    }
    u32 u[36];
#pragma unroll
    for (int i = 0; i < 16; i++) u[i] = cube(add(s[i], K.rc_init[3][i]));
    sfor<0, 20>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const u32* row = MODE == 1 ? kLinC.y[r] : MODE == 2 ? kLinL.y[r] : lds->y[r];
        u32 y = add(dot_n<16 + r>(u, row), row[36]);
        u[16 + r] = cube(y);
    });
    sfor<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const u32* row = MODE == 1 ? kLinC.fin[i] : MODE == 2 ? kLinL.fin[i] : lds->fin[i];
        s[i] = add(dot_n<36>(u, row), row[36]);
    });
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], K.rc_term[r][i]));
        mds_circ16(s);
    }
}

template <int V>
__global__ __launch_bounds__(256) void k_perm(u32* out, u32 seed, int reps) {
    __shared__ Lin lds;
    if (V == 3) {
        const u32* src = (const u32*)&kLinC;
        u32* dst = (u32*)&lds;
        for (u32 i = threadIdx.x; i < sizeof(Lin) / 4; i += 256) dst[i] = src[i];
        __syncthreads();
    }
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = (seed * 2654435761u + threadIdx.x * 16 + i + blockIdx.x * 7919u) % P;
    for (int r = 0; r < reps; r++) {
        if (V == 0) poseidon16_permute(s);
        else perm_lin<V>(s, &lds);
    }
    u32 x = 0;
    for (int i = 0; i < 16; i++) x ^= s[i] + i;
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int V>
static void run(u32* d_out, u32* h_ref, u32 n, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_perm<V><<<n / 256, 256>>>(d_out, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_perm<V><<<n / 256, 256>>>(d_out, 1, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    u32* h = (u32*)malloc(n * 4);
    hipMemcpy(h, d_out, n * 4, hipMemcpyDeviceToHost);
    u32 bad = 0;
    if (V == 0) memcpy(h_ref, h, n * 4); else for (u32 i = 0; i < n; i++) bad += h[i] != h_ref[i];
    printf("variant %d: %.3f ms  %.2f G perm/s  mismatches vs V0: %u\n", V, ms, (double)n * reps / ms * 1e-6, bad);
    free(h);
}
int main() {
    const u32 n = 256 * 256 * 16;
    const int reps = 16;
    u32* d; hipMalloc(&d, n * 4);
    u32* ref = (u32*)malloc(n * 4);
    run<0>(d, ref, n, reps); run<1>(d, ref, n, reps); run<2>(d, ref, n, reps); run<3>(d, ref, n, reps);
    return 0;
}
