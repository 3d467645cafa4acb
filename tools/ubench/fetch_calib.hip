// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses (MI355X_MICROARCH.md: the x2
// factor is documented for 16 B / lane streams only).  Every kernel reads (or writes) a known number of bytes from a 1 GiB buffer —
// beyond the 256 MiB Infinity Cache —: 4, 8 and 16 bytes per lane, consecutive lanes on consecutive addresses, plus the AIR
// kernels' pattern (uint2 per lane from 5 planes 64 MiB apart).
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- ./fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- ./fetch_calib
// tools/fetch_calib_summary.py prints counter / bytes per kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
typedef uint64_t u64;
static constexpr u64 WORDS = 1ull << 28;  // 1 GiB

__global__ __launch_bounds__(256) void read4(const u32* __restrict__ p, u32* out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < WORDS; i += (u64)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void read8(const uint2* __restrict__ p, u32* out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < WORDS / 2; i += (u64)gridDim.x * 256) {
        const uint2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void read16(const uint4* __restrict__ p, u32* out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < WORDS / 4; i += (u64)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the extension-field column read of k_air_round / k_gkr_step: one uint2 per lane from each of 5 planes
__global__ __launch_bounds__(256) void read8_planes5(const u32* __restrict__ p, u32* out) {
    const u64 plane = WORDS / 5 / 2 * 2;
    u32 acc = 0;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < plane / 2; j += (u64)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint2 v = *reinterpret_cast<const uint2*>(p + k * plane + 2 * j);
            acc ^= v.x ^ v.y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void write4(u32* __restrict__ p) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < WORDS; i += (u64)gridDim.x * 256) p[i] = (u32)i;
}
__global__ __launch_bounds__(256) void write16(uint4* __restrict__ p) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < WORDS / 4; i += (u64)gridDim.x * 256) p[i] = make_uint4((u32)i, 1, 2, 3);
}

int main() {
    u32 *d, *out;
    if (hipMalloc(&d, WORDS * 4) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(d, 1, WORDS * 4);
    hipDeviceSynchronize();
    const dim3 g(8192), b(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read4, g, b, 0, 0, d, out);
        hipLaunchKernelGGL(read8, g, b, 0, 0, (const uint2*)d, out);
        hipLaunchKernelGGL(read16, g, b, 0, 0, (const uint4*)d, out);
        hipLaunchKernelGGL(read8_planes5, g, b, 0, 0, d, out);
        hipLaunchKernelGGL(write4, g, b, 0, 0, d);
        hipLaunchKernelGGL(write16, g, b, 0, 0, (uint4*)d);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: read4 read8 read16 write4 write16 = %llu; read8_planes5 = %llu\n", (unsigned long long)(WORDS * 4),
           (unsigned long long)(WORDS / 5 / 2 * 2 * 5 * 4));
    return 0;
}
