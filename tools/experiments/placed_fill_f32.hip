// Round 6 experiment: the bare fp32 store pattern (15 planes of floats [n][runs], 256 B per store instruction) on placed memory, by
// how many wavefronts per CU issue the stores: 4 (what the fp32 kernel's consumers are), 8, 12, 16 (steps dealt round-robin to the
// wavefronts of a run group).  Is the fp32 launch (0.74 ms) at what its store concurrency allows?
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -o tools/build/placed_fill_f32 tools/experiments/placed_fill_f32.hip -Lgnss-ins-sim_amd/lib -lginsim
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "ginsim.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { printf("FAILED %s -> %d %s\n", #x, r_, ginsim_last_error()); exit(2); } } while (0)

// blockDim = 256 * K: thread t serves run (block * 256 + t % 256) and the steps j with j % K == t / 256
template <typename T, int K>
__global__ void __launch_bounds__(256 * K) fill(T* base, int64_t n, int64_t runs) {
    const int64_t r = (int64_t)blockIdx.x * 256 + (threadIdx.x & 255);
    const int g = threadIdx.x >> 8;
    const int64_t plane = n * runs;
    T v = (T)r;
    for (int64_t j = g; j < n; j += K) {
        v = v * (T)1.0000001 + (T)0.5;
#pragma unroll
        for (int c = 0; c < 15; ++c) __builtin_nontemporal_store(v + (T)c, base + c * plane + j * runs + r);
    }
}
static hipEvent_t e0, e1;
template <class F> static float avg_ms(F f, int warm, int reps) {
    for (int i = 0; i < warm; ++i) f();
    float s = 0;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); s += t; }
    return s / reps;
}
template <typename T> static void run_all(const char* tag, void* p, int64_t n, int64_t runs) {
    const double bytes = 15.0 * n * runs * sizeof(T);
    float a;
    a = avg_ms([&] { hipLaunchKernelGGL((fill<T, 1>), dim3(runs / 256), dim3(256), 0, 0, (T*)p, n, runs); }, 20, 20);
    printf("  %s  4 storing wavefronts per workgroup : %.4f ms = %.0f GB/s (%.3f)\n", tag, a, bytes / a / 1e6, bytes / a / 8e9);
    a = avg_ms([&] { hipLaunchKernelGGL((fill<T, 2>), dim3(runs / 256), dim3(512), 0, 0, (T*)p, n, runs); }, 20, 20);
    printf("  %s  8                                  : %.4f ms = %.0f GB/s (%.3f)\n", tag, a, bytes / a / 1e6, bytes / a / 8e9);
    a = avg_ms([&] { hipLaunchKernelGGL((fill<T, 3>), dim3(runs / 256), dim3(768), 0, 0, (T*)p, n, runs); }, 20, 20);
    printf("  %s 12                                  : %.4f ms = %.0f GB/s (%.3f)\n", tag, a, bytes / a / 1e6, bytes / a / 8e9);
    a = avg_ms([&] { hipLaunchKernelGGL((fill<T, 4>), dim3(runs / 256), dim3(1024), 0, 0, (T*)p, n, runs); }, 20, 20);
    printf("  %s 16                                  : %.4f ms = %.0f GB/s (%.3f)\n", tag, a, bytes / a / 1e6, bytes / a / 8e9);
}
int main() {
    CK(hipSetDevice(0)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ginsim_ctx* ctx; GK(ginsim_create(0, &ctx));
    const int64_t n = 1000;
    for (int64_t runs : {65536ll, 262144ll}) {
        const size_t total = (size_t)15 * n * runs * 8;
        void *placed = nullptr, *plain = nullptr;
        GK(ginsim_placed_reserve(ctx, total + (64u << 20)));
        GK(ginsim_malloc_placed(ctx, total, &placed));
        CK(hipMalloc(&plain, total));
        printf("runs %lld\n", (long long)runs);
        run_all<float>("f32 placed", placed, n, runs);
        run_all<float>("f32 plain ", plain, n, runs);
        run_all<double>("f64 placed", placed, n, runs);
        CK(hipFree(plain)); GK(ginsim_free(ctx, placed));
    }
    GK(ginsim_destroy(ctx));
    return 0;
}
