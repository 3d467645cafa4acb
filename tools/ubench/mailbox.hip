// micro-benchmark: round trip host <-> resident kernel through pinned coherent host memory, against the launch-per-exchange
// pattern (one tiny kernel that publishes a flag, host spins on it).  Decides whether a sumcheck tail that stays resident
// on the device and receives its challenges through a mailbox beats one launch per two rounds.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <csignal>
#include <unistd.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;

__device__ __forceinline__ void store_sys(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u32 load_sys(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// resident: n exchanges inside one launch.  payload: 64 words up, 16 words down.
__global__ __launch_bounds__(1024) void k_resident(u32* up, u32* up_flag, const u32* down, const u32* down_flag, u32 n, u32* err) {
    __shared__ u32 sh[16];
    u32 x = threadIdx.x;
    for (u32 it = 1; it <= n; it++) {
        if (threadIdx.x < 64) {
            store_sys(up + threadIdx.x, x + it);
            __builtin_amdgcn_s_waitcnt(0);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            store_sys(up_flag, it);
            const unsigned long long t0 = wall_clock64();
            while (load_sys(down_flag) != it) {
                if (wall_clock64() - t0 > 200000000ull) {  // 2 s at 100 MHz: give up
                    store_sys(err, 1);
                    break;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < 16) sh[threadIdx.x] = load_sys(down + threadIdx.x);
        __syncthreads();
        x += sh[threadIdx.x & 15];
    }
    if (threadIdx.x == 0) store_sys(up + 100, x);
}
__global__ __launch_bounds__(1024) void k_one(u32* up, u32* up_flag, const u32* down, u32 it) {
    u32 x = threadIdx.x + down[threadIdx.x & 15];
    if (threadIdx.x < 64) {
        store_sys(up + threadIdx.x, x + it);
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    if (threadIdx.x == 0) store_sys(up_flag, it);
}

int main() {
    u32* h;  // [0..128) up payload, [128] up flag, [192..208) down payload, [256] down flag, [320] err
    CHECK(hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < 1024; i++) h[i] = 0;
    volatile u32* up_flag = h + 128;
    volatile u32* down_flag = h + 256;
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const u32 n = 2000;
    for (int rep = 0; rep < 3; rep++) {
        *up_flag = 0;
        *down_flag = 0;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_resident, dim3(1), dim3(1024), 0, s, h, h + 128, h + 192, h + 256, n, h + 320);
        for (u32 it = 1; it <= n; it++) {
            while (*up_flag != it) __builtin_ia32_pause();
            for (int i = 0; i < 16; i++) h[192 + i] = h[i] ^ it;  // "transcript": read the payload, write the answer
            __atomic_thread_fence(__ATOMIC_RELEASE);
            *down_flag = it;
        }
        CHECK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        printf("resident: %.2f us per exchange (err %u)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n, h[320]);
    }
    for (int rep = 0; rep < 3; rep++) {
        *up_flag = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (u32 it = 1; it <= n; it++) {
            hipLaunchKernelGGL(k_one, dim3(1), dim3(1024), 0, s, h, h + 128, h + 192, it);
            while (*up_flag != it) __builtin_ia32_pause();
            for (int i = 0; i < 16; i++) h[192 + i] = h[i] ^ it;
        }
        CHECK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        printf("launch per exchange: %.2f us per exchange\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    }
    // the same with the down mailbox in fine-grained DEVICE memory written by the host through the BAR (if the runtime allows it):
    // the kernel then polls its own HBM instead of reading host memory over PCIe, the host's message is a posted write
    u32* dmb = nullptr;
    if (hipExtMallocWithFlags((void**)&dmb, 4096, hipDeviceMallocFinegrained) == hipSuccess && dmb) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, dmb) == hipSuccess) printf("fine-grained device memory: host pointer %p device pointer %p\n", a.hostPointer, a.devicePointer);
        CHECK(hipMemset(dmb, 0, 4096));
        CHECK(hipDeviceSynchronize());
        signal(SIGSEGV, [](int) { const char m[] = "host write to fine-grained device memory: SIGSEGV (not host-accessible)\n"; (void)!write(1, m, sizeof m - 1); _exit(0); });
        signal(SIGBUS, [](int) { const char m[] = "host write to fine-grained device memory: SIGBUS\n"; (void)!write(1, m, sizeof m - 1); _exit(0); });
        volatile u32* d = dmb;
        d[64] = 7;  // probe
        printf("host write to fine-grained device memory works (read back %u)\n", d[64]);
        for (int rep = 0; rep < 3; rep++) {
            *up_flag = 0;
            d[256] = 0;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_resident, dim3(1), dim3(1024), 0, s, h, h + 128, (const u32*)dmb + 192, (const u32*)dmb + 256, n, h + 320);
            for (u32 it = 1; it <= n; it++) {
                while (*up_flag != it) __builtin_ia32_pause();
                for (int i = 0; i < 16; i++) d[192 + i] = h[i] ^ it;
                __atomic_thread_fence(__ATOMIC_RELEASE);
                d[256] = it;
            }
            CHECK(hipStreamSynchronize(s));
            auto t1 = std::chrono::steady_clock::now();
            printf("resident, mailbox in device memory: %.2f us per exchange (err %u)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n, h[320]);
        }
    } else {
        printf("hipExtMallocWithFlags(finegrained) failed\n");
    }
    return 0;
}
