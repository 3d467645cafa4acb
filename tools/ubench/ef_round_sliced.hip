// micro-benchmark: one "EF full round" (S-box x^3 on 16 extension-field elements + circulant MDS per coefficient plane),
// the inner loop of the Poseidon AIR in extension-field sumcheck rounds.
//   A  monolithic: one lane holds the whole state (16 x 5 words = 80 VGPRs + temporaries; 1 wave/SIMD in the product)
//   B  plane-sliced: five adjacent lanes hold one coefficient plane each (16 words); EF multiplications gather the other
//      planes with ds_bpermute; MDS is lane-local.  3 groups of 5 per DPP row of 16 (lanes 15, 31, 47, 63 idle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using namespace kb;

__device__ __forceinline__ EF cube_ef(const EF& a) { return ef_mul(ef_mul(a, a), a); }
__global__ __launch_bounds__(256, 1) void k_mono(u32* out, u32 seed, int reps) {
    EF s[16];
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 5; k++) s[i].v[k] = (seed * 2654435761u + (blockIdx.x * 256 + threadIdx.x) * 80 + i * 5 + k) % P;
    for (int r = 0; r < reps; r++) {
        static_for<0, 16>([&](auto I) { constexpr int i = decltype(I)::value; s[i] = cube_ef(ef_add_base(s[i], 12345 + i)); });
#pragma unroll
        for (int k = 0; k < 5; k++) {
            u32 p[16];
#pragma unroll
            for (int i = 0; i < 16; i++) p[i] = s[i].v[k];
            mds_circ16(p);
#pragma unroll
            for (int i = 0; i < 16; i++) s[i].v[k] = p[i];
        }
    }
    u32 x = 0;
    for (int i = 0; i < 16; i++) for (int k = 0; k < 5; k++) x ^= s[i].v[k] + i * 5 + k;
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

// ---- sliced ----
struct Slice {
    u32 k;          // my plane
    int src[5];     // byte address (lane*4) of the lane holding plane j of my group
    // row k of the multiplication matrix M(b): entry i = cand[sel[i]] with cand = {b0,b1,b2,b3,b4,b0m3,b1m4,b4m2,b3m14}
    u32 sel[5];
};
__device__ __forceinline__ void gather(u32 mine, const Slice& S, u32 all[5]) {
#pragma unroll
    for (int j = 0; j < 5; j++) all[j] = (u32)__builtin_amdgcn_ds_bpermute(S.src[j], (int)mine);
}
// plane k of a*b given all planes of both (every lane computes only ITS row)
__device__ __forceinline__ u32 mul_plane(const u32 a[5], const u32 b[5], const Slice& S) {
    u32 cand[9] = {b[0], b[1], b[2], b[3], b[4], sub(b[0], b[3]), sub(b[1], b[4]), sub(b[4], b[2]), 0};
    cand[8] = sub(b[3], cand[6]);
    u32 row[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u32 v = cand[0];
#pragma unroll
        for (int c = 1; c < 9; c++) v = S.sel[i] == (u32)c ? cand[c] : v;
        row[i] = v;
    }
    return dot5(a, row[0], row[1], row[2], row[3], row[4]);
}
__global__ __launch_bounds__(256) void k_sliced(u32* out, u32 seed, int reps) {
    const u32 lane = threadIdx.x & 63, in_row = lane & 15, grp = in_row / 5, k = in_row % 5;
    const bool idle = in_row == 15;
    Slice S;
    S.k = k;
    for (int j = 0; j < 5; j++) S.src[j] = (int)(((lane & ~15u) + grp * 5 + j) * 4);
    static const u32 SEL[5][5] = {{0, 4, 3, 2, 6}, {1, 0, 4, 3, 2}, {2, 6, 5, 7, 8}, {3, 2, 6, 5, 7}, {4, 3, 2, 6, 5}};
    for (int i = 0; i < 5; i++) S.sel[i] = idle ? 0 : SEL[k][i];
    const u32 gid = (blockIdx.x * 256 + threadIdx.x) / 16 * 3 + grp;  // state index
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = (seed * 2654435761u + gid * 80 + i * 5 + k) % P;
    for (int r = 0; r < reps; r++) {
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const u32 x = k == 0 ? add(s[i], 12345 + i) : s[i];
            u32 a[5];
            gather(x, S, a);
            const u32 sq = mul_plane(a, a, S);
            u32 q[5];
            gather(sq, S, q);
            s[i] = mul_plane(q, a, S);
        });
        mds_circ16(s);
    }
    u32 x = 0;
    for (int i = 0; i < 16; i++) x ^= s[i] + i * 5 + k;
    // combine the group's 5 planes so that the result is comparable with the monolithic kernel
    u32 all[5];
    gather(x, S, all);
    if (!idle && k == 0) out[gid] = all[0] ^ all[1] ^ all[2] ^ all[3] ^ all[4];
}
int main() {
    const u32 n_states = 256 * 256 * 12;  // multiple of 48 (sliced: 12 states per wave... 48 per block)
    const int reps = 8;
    u32 *d1, *d2;
    hipMalloc(&d1, n_states * 4);
    hipMalloc(&d2, n_states * 4 + 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    // monolithic: one state per lane
    k_mono<<<n_states / 256, 256>>>(d1, 1, 1);
    hipDeviceSynchronize();
    hipEventRecord(a); k_mono<<<n_states / 256, 256>>>(d1, 1, reps); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("monolithic: %.3f ms  %.2f G EF-rounds/s\n", ms, (double)n_states * reps / ms * 1e-6);
    // sliced: 48 states per block of 256 lanes
    const u32 blocks = n_states / 48;
    k_sliced<<<blocks, 256>>>(d2, 1, 1);
    hipDeviceSynchronize();
    hipEventRecord(a); k_sliced<<<blocks, 256>>>(d2, 1, reps); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("sliced:     %.3f ms  %.2f G EF-rounds/s\n", ms, (double)n_states * reps / ms * 1e-6);
    u32* h1 = (u32*)malloc(n_states * 4); u32* h2 = (u32*)malloc(n_states * 4);
    hipMemcpy(h1, d1, n_states * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n_states * 4, hipMemcpyDeviceToHost);
    // same seeds per state? monolithic state index = global thread id; sliced gid enumerates the same range
    u32 bad = 0;
    for (u32 i = 0; i < n_states; i++) bad += h1[i] != h2[i];
    printf("mismatches: %u of %u\n", bad, n_states);
    return 0;
}
