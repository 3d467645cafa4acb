// micro-benchmark: one kernel reading S streams (the 4 quarters x 5 planes of W + 4 quarters of f in k_prod_round2 are 24) whose
// bases are a power of two apart, against the same streams skewed by a few hundred bytes each: does the address pattern of the
// MSB-first product sumcheck cost HBM bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
typedef unsigned long long u64;

template <int S, int VEC>
__global__ __launch_bounds__(256) void k_read(const u32* __restrict__ base, u64 stride_words, u64 n, u32* __restrict__ out) {
    u32 acc = 0;
    for (u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += (u64)gridDim.x * 256 * VEC) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            const u32* p = base + (u64)s * stride_words + i;
            if (VEC == 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else {
                acc += *p;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int S, int VEC>
float run(const u32* d, u64 stride, u64 n, u32* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_read<S, VEC>), dim3(blocks), dim3(256), 0, 0, d, stride, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const u64 n = 1ull << 24;            // words per stream (64 MB), as the quarters of the first sumcheck pass
    const int S = 24;
    const u64 total = (u64)S * (n + 4096) + 4096;
    u32 *d, *out;
    CHECK(hipMalloc(&d, total * 4));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(d, 1, total * 4));
    const double gb = (double)S * n * 4 / 1e9;
    for (int blocks : {2048, 4096, 8192}) {
        const float a = run<24, 1>(d, n, n, out, blocks), b = run<24, 1>(d, n + 96, n, out, blocks), c = run<24, 1>(d, n + 1056, n, out, blocks);
        const float a4 = run<24, 4>(d, n, n, out, blocks), b4 = run<24, 4>(d, n + 96, n, out, blocks), c4 = run<24, 4>(d, n + 1056, n, out, blocks);
        printf("blocks %5d  dword:   pow2 stride %.3f ms (%.2f TB/s)  +384 B %.3f ms (%.2f TB/s)  +4224 B %.3f ms (%.2f TB/s)\n", blocks, a, gb / a, b, gb / b, c, gb / c);
        printf("blocks %5d  dwordx4: pow2 stride %.3f ms (%.2f TB/s)  +384 B %.3f ms (%.2f TB/s)  +4224 B %.3f ms (%.2f TB/s)\n", blocks, a4, gb / a4, b4, gb / b4, c4, gb / c4);
    }
    const float one = run<1, 4>(d, 0, (u64)S * n, out, 8192);
    printf("one stream of the same size, dwordx4: %.3f ms (%.2f TB/s)\n", one, gb / one);
    return 0;
}
