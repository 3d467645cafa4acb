// micro-benchmark: integer instruction and Poseidon throughput on gfx950 (chip-wide, lane-ops per second)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using kb::u32; using kb::u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k_rate(u32* out, u32 seed) {
    u32 a[8];
    u64 w[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 8 + i; w[i] = a[i]; }
    u32 c = seed | 1;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) w[i] = (u64)(u32)w[i] * c + w[i];                    // v_mad_u64_u32
            // (as C, `a *= c` 4096 times is folded into one multiplication by c^4096 — round 2's "270 T/s"; the instruction is
            // pinned with inline assembly, one dependent v_mul_lo_u32 per iteration and chain)
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));   // v_mul_lo_u32
            if (OP == 2) a[i] = __umulhi(a[i], c) + 1;                         // v_mul_hi_u32
            if (OP == 3) a[i] = __umul24(a[i], c) + a[i];                      // v_mad_u32_u24
            if (OP == 4) a[i] = kb::mul(a[i] & 0x3fffffff, c & 0x3fffffff);   // Montgomery mul
            if (OP == 5) a[i] = kb::add(a[i] & 0x3fffffff, c & 0x3fffffff);   // modular add
            if (OP == 6) w[i] = (w[i] << 3) + w[i];                            // v_lshl_add_u64
            if (OP == 7) a[i] = (a[i] << 3) + c;                               // v_lshl_add_u32
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (u32)w[i] + (u32)(w[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_perm(u32* out, u32 seed, int reps) {
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = (seed + threadIdx.x * 16 + i) & 0x3fffffff;
    for (int r = 0; r < reps; r++) kb::poseidon16_permute(s);
    u32 x = 0;
    for (int i = 0; i < 16; i++) x += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_mds(u32* out, u32 seed, int reps) {
    u32 s[16];
    for (int i = 0; i < 16; i++) s[i] = (seed + threadIdx.x * 16 + i) & 0x3fffffff;
    for (int r = 0; r < reps; r++) kb::mds_circ16(s);
    u32 x = 0;
    for (int i = 0; i < 16; i++) x += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
int main() {
    u32* d; CHECK(hipMalloc(&d, 8192 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 8192;
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32(+add)", "v_mad_u32_u24", "kb::mul", "kb::add", "v_lshl_add_u64", "v_lshl_add_u32"};
    auto run = [&](int op) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (op) {
                case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 5: hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 6: hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
                case 7: hipLaunchKernelGGL(k_rate<7>, dim3(blocks), dim3(256), 0, 0, d, 12345u); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)blocks * 256 * 8 * ITERS;
        printf("%-20s %8.3f ms  %7.2f T lane-ops/s\n", names[op], ms, ops / ms / 1e9);
    };
    for (int op = 0; op < 8; op++) run(op);
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, d, 7u, 8); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("poseidon16_permute   %8.3f ms  %7.2f G perm/s\n", ms, (double)blocks * 256 * 8 / ms / 1e6);
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); hipLaunchKernelGGL(k_mds, dim3(blocks), dim3(256), 0, 0, d, 7u, 64); hipEventRecord(e1); hipEventSynchronize(e1); }
    hipEventElapsedTime(&ms, e0, e1);
    printf("mds_circ16           %8.3f ms  %7.2f G mds/s\n", ms, (double)blocks * 256 * 64 / ms / 1e6);
    return 0;
}
