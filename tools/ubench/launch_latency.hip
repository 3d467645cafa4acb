// micro-benchmark: launch -> host-visible flag latency (floor of one Fiat-Shamir round trip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_flag(unsigned* flag, unsigned seq) {
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
__global__ void k_work(unsigned* flag, unsigned seq, const unsigned* data, unsigned* out) {
    __shared__ unsigned l[256];
    unsigned v = data[threadIdx.x];
    for (int i = 0; i < 200; i++) v = v * 1664525u + 1013904223u;
    l[threadIdx.x] = v; __syncthreads();
    if (threadIdx.x == 0) { unsigned s = 0; for (int i = 0; i < 256; i++) s += l[i]; out[0] = s; __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* h; hipHostMalloc((void**)&h, 64, hipHostMallocMapped | hipHostMallocCoherent); h[0] = 0;
    unsigned *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 64); hipMemset(d, 1, 4096);
    for (int mode = 0; mode < 3; mode++) {
        const int N = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= N; i++) {
            unsigned seq = mode * 100000 + i;
            if (mode == 0) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, h, seq); volatile unsigned* f = h; while (*f != seq) {} }
            if (mode == 1) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, h, seq); hipStreamSynchronize(st); }
            if (mode == 2) { hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, h, seq, d, o); volatile unsigned* f = h; while (*f != seq) {} }
        }
        auto t1 = std::chrono::steady_clock::now();
        printf("mode %d (%s): %.2f us per round trip\n", mode, mode == 0 ? "flag poll" : mode == 1 ? "stream sync" : "small work + flag poll",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
}
