// Round 6 experiment: the bare READ 6 planes + WRITE 9 planes pattern of the given-sensors mechanisation (48 B read, 72 B written per
// sample and run) on placed and on plain memory, with 1, 2, 4 wavefronts per SIMD and with the loads of the next step(s) issued ahead.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "ginsim.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { printf("FAILED %s -> %d %s\n", #x, r_, ginsim_last_error()); exit(2); } } while (0)

// AHEAD: how many steps ahead the six loads are issued (a register ring of AHEAD + 1 steps)
template <int AHEAD>
__global__ void __launch_bounds__(256) mixed(const double* in, double* out, int64_t n, int64_t runs) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = n * runs;
    double buf[AHEAD + 1][6];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) buf[a][c] = __builtin_nontemporal_load(in + c * plane + (int64_t)a * runs + r);
    double acc = 0.0;
    for (int64_t j = 0; j < n; j += AHEAD + 1) {
#pragma unroll
        for (int u = 0; u <= AHEAD; ++u) {
            const int64_t jj = j + u;
            if (jj + AHEAD < n) {
#pragma unroll
                for (int c = 0; c < 6; ++c) buf[(u + AHEAD) % (AHEAD + 1)][c] = __builtin_nontemporal_load(in + c * plane + (jj + AHEAD) * runs + r);
            }
            if (jj < n) {
                double s = acc;
#pragma unroll
                for (int c = 0; c < 6; ++c) s += buf[u][c];
                acc = s * 0.5;
#pragma unroll
                for (int c = 0; c < 9; ++c) __builtin_nontemporal_store(s + c, out + c * plane + jj * runs + r);
            }
        }
    }
}
static hipEvent_t e0, e1;
template <class F> static float avg_ms(F f, int warm, int reps) {
    for (int i = 0; i < warm; ++i) f();
    float s = 0;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); s += t; }
    return s / reps;
}
int main() {
    CK(hipSetDevice(0)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ginsim_ctx* ctx; GK(ginsim_create(0, &ctx));
    const int64_t n = 1000;
    for (int64_t runs : {65536ll, 262144ll}) {
        const size_t plane = (size_t)n * runs * 8;
        void *pin = nullptr, *pout = nullptr, *hin = nullptr, *hout = nullptr;
        GK(ginsim_placed_reserve(ctx, 15 * plane + (64u << 20)));
        GK(ginsim_malloc_placed(ctx, 6 * plane, &pin)); GK(ginsim_malloc_placed(ctx, 9 * plane, &pout));
        CK(hipMalloc(&hin, 6 * plane)); CK(hipMalloc(&hout, 9 * plane));
        CK(hipMemset(pin, 0, 6 * plane)); CK(hipMemset(hin, 0, 6 * plane));
        const double bytes = 15.0 * plane;
        printf("runs %lld (%.0f wavefronts per SIMD)\n", (long long)runs, runs / 65536.0);
        for (int which = 0; which < 2; ++which) {
            const double* i = (const double*)(which ? pin : hin); double* o = (double*)(which ? pout : hout);
            const dim3 g(runs / 256), b(256);
            float t0 = avg_ms([&] { hipLaunchKernelGGL((mixed<0>), g, b, 0, 0, i, o, n, runs); }, 10, 15);
            float t1 = avg_ms([&] { hipLaunchKernelGGL((mixed<1>), g, b, 0, 0, i, o, n, runs); }, 10, 15);
            float t3 = avg_ms([&] { hipLaunchKernelGGL((mixed<3>), g, b, 0, 0, i, o, n, runs); }, 10, 15);
            float t7 = avg_ms([&] { hipLaunchKernelGGL((mixed<7>), g, b, 0, 0, i, o, n, runs); }, 10, 15);
            printf("  %-6s loads 0 / 1 / 3 / 7 steps ahead: %.4f / %.4f / %.4f / %.4f ms = %.3f / %.3f / %.3f / %.3f of 8 TB/s\n", which ? "placed" : "plain",
                   t0, t1, t3, t7, bytes / t0 / 8e9, bytes / t1 / 8e9, bytes / t3 / 8e9, bytes / t7 / 8e9);
        }
        CK(hipFree(hin)); CK(hipFree(hout)); GK(ginsim_free(ctx, pin)); GK(ginsim_free(ctx, pout));
    }
    GK(ginsim_destroy(ctx));
    return 0;
}
