// A FOREIGN tenant for the stress leg of the resident kernels (tools/stress_inflight.py under `hog`): a plain persistent HIP kernel
// that knows nothing of this library — it is not registered in the provers' shared counter — and holds wave slots for a while.
// usage: ./hog [workgroups=256] [seconds=30] [threads=1024]
//   256 workgroups of 1024 threads = 16 waves (4 per SIMD) on every CU of an MI355X, for the whole time: a 1024-thread workgroup of a
//   resident GKR tail (> 64 VGPRs) finds no CU with room for it until the hog leaves — the starvation the fail-soft path is for.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_hog(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        x = x * 1664525u + 1013904223u;
        __builtin_amdgcn_s_sleep(8);
    }
    if (x == 0xdeadbeefu) *sink = x;
}

// the same with ~120 live VGPRs per lane: 1024 threads x 120 registers fill the register file of a CU — NOTHING else fits beside
// such a workgroup (usage: 4th argument = 1).  248 of them leave 8 of the 256 CUs to everybody else.
__global__ __launch_bounds__(1024) void k_hog_heavy(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned r[112];
#pragma unroll
    for (int i = 0; i < 112; i++) r[i] = threadIdx.x * 2654435761u + i;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 112; i++) r[i] = r[i] * 1664525u + r[(i + 1) % 112];
        __builtin_amdgcn_s_sleep(8);
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 112; i++) x ^= r[i];
    if (x == 0xdeadbeefu) *sink = x;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const double seconds = argc > 2 ? atof(argv[2]) : 30.0;
    const int threads = argc > 3 ? atoi(argv[3]) : 1024;
    unsigned* sink;
    if (hipMalloc(&sink, 4) != hipSuccess) return 1;
    // one launch per <= 2 s (a kernel that runs for minutes looks like a hang to watchdogs); back to back, so the slots are never free for long
    double left = seconds;
    while (left > 0) {
        const double s = left > 2.0 ? 2.0 : left;
        if (argc > 4 && atoi(argv[4]) == 1)
            hipLaunchKernelGGL(k_hog_heavy, dim3(wgs), dim3(1024), 0, 0, (unsigned long long)(s * 1e8), sink);
        else
            hipLaunchKernelGGL(k_hog, dim3(wgs), dim3(threads), 0, 0, (unsigned long long)(s * 1e8), sink);  // wall_clock64: 100 MHz
        left -= s;
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("hog: %d workgroups x %d threads for %.1f s done\n", wgs, threads, seconds);
    return 0;
}
