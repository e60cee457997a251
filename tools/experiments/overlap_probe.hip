// Round 6 experiment: how well do a wavefront's store bursts overlap its own arithmetic?  One lane = one run, time loop inside, 15
// planes [n][runs] on placed memory; per step W dependent fmas (a stand-in for the mechanisation) and / or the 15 stores.
//   stores only | arithmetic only | both     -> is "both" the maximum of the two, or their sum?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "ginsim.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { printf("FAILED %s -> %d %s\n", #x, r_, ginsim_last_error()); exit(2); } } while (0)

// ILP independent chains of W / ILP fmas each: the arithmetic of a step (W fmas in all)
template <typename T, bool STORE, int ILP>
__global__ void __launch_bounds__(256) k(T* base, int64_t n, int64_t runs, int W, T* sink) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n * runs;
    T v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = (T)(r + i);
    for (int64_t j = 0; j < n; ++j) {
        for (int w = 0; w < W / ILP; ++w)
#pragma unroll
            for (int i = 0; i < ILP; ++i) v[i] = v[i] * (T)1.0000001 + (T)0.5;
        if (STORE) {
#pragma unroll
            for (int c = 0; c < 15; ++c) __builtin_nontemporal_store(v[c % ILP] + (T)c, base + c * plane + j * runs + r);
        }
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    if (s == (T)1.2345e-30) sink[0] = s;
}
// The same step with the stores handed to ANOTHER wavefront through LDS: threads 0..255 do the arithmetic and drop the 15 values of
// their run into a double-buffered queue, threads 256..511 pick up the previous step's values and issue the stores; one barrier
// per step.  A store that waits for room in the CU's queue then stalls a wavefront that has nothing else to do.
template <typename T, int ILP>
__global__ void __launch_bounds__(512) k2(T* base, int64_t n, int64_t runs, int W, T* sink) {
    extern __shared__ double lds_raw[];
    T* q = reinterpret_cast<T*>(lds_raw);                 // [2][15][256]
    const int lane = threadIdx.x & 255;
    const bool storer = threadIdx.x >= 256;
    const int64_t r = (int64_t)blockIdx.x * 256 + lane;
    const int64_t plane = n * runs;
    T v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = (T)(r + i);
    for (int64_t j = 0; j <= n; ++j) {
        if (!storer) {
            if (j < n) {
                for (int w = 0; w < W / ILP; ++w)
#pragma unroll
                    for (int i = 0; i < ILP; ++i) v[i] = v[i] * (T)1.0000001 + (T)0.5;
                T* slot = q + (j & 1) * 15 * 256 + lane;
#pragma unroll
                for (int c = 0; c < 15; ++c) slot[c * 256] = v[c % ILP] + (T)c;
            }
        } else if (j >= 1) {
            const T* slot = q + ((j - 1) & 1) * 15 * 256 + lane;
#pragma unroll
            for (int c = 0; c < 15; ++c) __builtin_nontemporal_store(slot[c * 256], base + c * plane + (j - 1) * runs + r);
        }
        __syncthreads();
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    if (s == (T)1.2345e-30) sink[0] = s;
}
static hipEvent_t e0, e1;
template <class F> static float avg_ms(F f, int warm, int reps) {
    for (int i = 0; i < warm; ++i) f();
    float s = 0;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); s += t; }
    return s / reps;
}
template <typename T> static void table(const char* tag, void* p, T* sink, int64_t n, int64_t runs) {
    const dim3 g(runs / 256), b(256);
    const float st = avg_ms([&] { hipLaunchKernelGGL((k<T, true, 4>), g, b, 0, 0, (T*)p, n, runs, 0, sink); }, 20, 15);
    printf("%s: stores only %.4f ms\n", tag, st);
    for (int W : {64, 128, 192, 256, 320, 384, 448, 512}) {
        const float c = avg_ms([&] { hipLaunchKernelGGL((k<T, false, 4>), g, b, 0, 0, (T*)p, n, runs, W, sink); }, 5, 8);
        const float both = avg_ms([&] { hipLaunchKernelGGL((k<T, true, 4>), g, b, 0, 0, (T*)p, n, runs, W, sink); }, 5, 8);
        const float handed = avg_ms([&] { hipLaunchKernelGGL((k2<T, 4>), g, dim3(512), 2 * 15 * 256 * sizeof(T), 0, (T*)p, n, runs, W, sink); }, 5, 8);
        printf("  W %3d fma/step (4 chains): arithmetic only %.4f  both %.4f   max %.4f  sum %.4f   both / max %.3f | stores handed to a second wavefront %.4f = %.3f x max\n", W, c, both, std::max(c, st), c + st, both / std::max(c, st), handed, handed / std::max(c, st));
    }
}
int main() {
    CK(hipSetDevice(0)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ginsim_ctx* ctx; GK(ginsim_create(0, &ctx));
    const int64_t n = 1000, runs = 65536;
    const size_t total = (size_t)15 * n * runs * 8;
    void* placed = nullptr; void* sink = nullptr;
    GK(ginsim_placed_reserve(ctx, total + (64u << 20)));
    GK(ginsim_malloc_placed(ctx, total, &placed));
    CK(hipMalloc(&sink, 64));
    table<double>("fp64 pattern, one wavefront per SIMD", placed, (double*)sink, n, runs);
    table<float>("fp32 pattern, one wavefront per SIMD", placed, (float*)sink, n, runs);
    GK(ginsim_free(ctx, placed)); GK(ginsim_destroy(ctx));
    return 0;
}
