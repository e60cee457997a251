// Round 6 experiment: the bare 15-plane store pattern (no arithmetic) on memory carved from the library's placed arena, against
// plain hipMalloc memory: is the fused kernel's 1.22 ms on placed planes the kernel's own limit or the layout's?
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -o tools/build/placed_fill tools/experiments/placed_fill.hip -Lgnss-ins-sim_amd/lib -lginsim -Wl,-rpath,$PWD/gnss-ins-sim_amd/lib
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "ginsim.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { printf("FAILED %s -> %d %s\n", #x, r_, ginsim_last_error()); exit(2); } } while (0)

template <int PLANES>
__global__ void __launch_bounds__(256) fill_planes(double* base, int64_t n, int64_t runs, int64_t plane_stride, int work = 0) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v = (double)r;
    for (int64_t j = 0; j < n; ++j) {
        v = v * 1.0000001 + 0.5;
        for (int w = 0; w < work; ++w) v = __builtin_fma(v, 1.0000001, 0.5);      // a dependent chain between the store bursts
#pragma unroll
        for (int c = 0; c < PLANES; ++c) __builtin_nontemporal_store(v + c, base + c * plane_stride + j * runs + r);
    }
}
static hipEvent_t e0, e1;
template <class F> static float avg_ms(F f, int warm, int reps, float* mn) {
    for (int i = 0; i < warm; ++i) f();
    float s = 0, b = 1e30f;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); s += t; b = std::min(b, t); }
    *mn = b; return s / reps;
}
int main() {
    CK(hipSetDevice(0)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ginsim_ctx* ctx; GK(ginsim_create(0, &ctx));
    const int64_t n = 1000;
    for (int64_t runs : {65536ll, 131072ll}) {
        const size_t plane = (size_t)n * runs * 8, total = 15 * plane;
        void *placed = nullptr, *plain = nullptr;
        GK(ginsim_placed_reserve(ctx, total + (64u << 20)));
        GK(ginsim_malloc_placed(ctx, total, &placed));
        CK(hipMalloc(&plain, total));
        ginsim_placed_info info; GK(ginsim_placed_info_get(ctx, &info));
        printf("runs %lld: arena %s, %lld stripes by class %lld %lld %lld, search %.2f s\n", (long long)runs, info.stripe_classes, (long long)(info.mapped_bytes / info.stripe_bytes),
               (long long)info.stripes_of_class[0], (long long)info.stripes_of_class[1], (long long)info.stripes_of_class[2], info.search_seconds);
        for (int rep = 0; rep < 2; ++rep) {
            for (int which = 0; which < 2; ++which) {
                double* p = (double*)(which ? placed : plain);
                float mn, a = avg_ms([&] { hipLaunchKernelGGL((fill_planes<15>), dim3(runs / 256), dim3(256), 0, 0, p, n, runs, (int64_t)(plane / 8)); }, 30, 30, &mn);
                printf("  %-7s 15 planes contiguous            : avg %.4f min %.4f ms = %.0f GB/s (%.3f of 8 TB/s)\n", which ? "placed" : "plain", a, mn, total / a / 1e6, total / a / 1e6 / 8000);
            }
        }
        for (int work : {50, 100, 150, 200, 250, 300}) {
            for (int which = 0; which < 2; ++which) {
                double* p = (double*)(which ? placed : plain);
                float mn, a = avg_ms([&] { hipLaunchKernelGGL((fill_planes<15>), dim3(runs / 256), dim3(256), 0, 0, p, n, runs, (int64_t)(plane / 8), work); }, 10, 20, &mn);
                printf("  %-7s 15 planes + %3d dependent fma/step : avg %.4f ms = %.0f GB/s\n", which ? "placed" : "plain", work, a, total / a / 1e6);
            }
        }
        // padded plane stride inside the placed region would need a larger region; instead: 12 planes with a stride of 1.25 planes (fits in 15)
        for (int which = 0; which < 2; ++which) {
            double* p = (double*)(which ? placed : plain);
            float mn, a = avg_ms([&] { hipLaunchKernelGGL((fill_planes<12>), dim3(runs / 256), dim3(256), 0, 0, p, n, runs, (int64_t)(plane / 8 * 5 / 4)); }, 30, 30, &mn);
            printf("  %-7s 12 planes, stride 1.25 planes    : avg %.4f min %.4f ms = %.0f GB/s\n", which ? "placed" : "plain", a, mn, 12.0 * plane / a / 1e6);
        }
        CK(hipFree(plain));
        GK(ginsim_free(ctx, placed));
    }
    GK(ginsim_destroy(ctx));
    return 0;
}
