// Do bursts of stores overlap with compute?  (development aid)  One lane = one run, per step K fp64 FMAs and 15
// 8-byte stores in the MC kernel's [plane][step][run] pattern; BURST: all 15 stores at the end of the step,
// SPREAD: one store after every K/15 FMAs.  Occupancy capped at 2 workgroups per CU with a dynamic-LDS reservation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int K>      // MODE 0: no stores, 1: burst, 2: spread
__global__ void __launch_bounds__(256) k(double* p, int64_t n, int64_t runs, double c) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = n * runs;
    double a[4] = {(double)r, r + 1.0, r + 2.0, r + 3.0};
    for (int64_t j = 0; j < n; ++j) {
        if (MODE == 2) {
#pragma unroll
            for (int s = 0; s < 15; ++s) {
#pragma unroll
                for (int i = 0; i < K / 15; ++i) a[i & 3] = __builtin_fma(a[i & 3], c, 1e-9);
                p[s * plane + j * runs + r] = a[s & 3];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 3) {          // spread, 8 x 16-byte stores ([pair][step][run][2] layout)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int i = 0; i < K / 8; ++i) a[i & 3] = __builtin_fma(a[i & 3], c, 1e-9);
                double2* q = reinterpret_cast<double2*>(p) + (s * plane + j * runs + r);
                *q = double2{a[s & 3], a[(s + 1) & 3]};
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 4) {          // spread, scalar base + 32-bit lane offset
            const uint32_t r32 = (uint32_t)r;
#pragma unroll
            for (int s = 0; s < 15; ++s) {
#pragma unroll
                for (int i = 0; i < K / 15; ++i) a[i & 3] = __builtin_fma(a[i & 3], c, 1e-9);
                double* base = p + (s * plane + j * runs);
                base[r32] = a[s & 3];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 5) {          // spread, 4-byte stores (fp32 pattern), 15 planes
#pragma unroll
            for (int s = 0; s < 15; ++s) {
#pragma unroll
                for (int i = 0; i < K / 15; ++i) a[i & 3] = __builtin_fma(a[i & 3], c, 1e-9);
                reinterpret_cast<float*>(p)[s * plane + j * runs + r] = (float)a[s & 3];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int o = 0; o < 15; ++o) {
#pragma unroll
                for (int i = 0; i < K / 15; ++i) a[i & 3] = __builtin_fma(a[i & 3], c, 1e-9);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1) {
#pragma unroll
                for (int s = 0; s < 15; ++s) p[s * plane + j * runs + r] = a[s & 3];
            }
        }
    }
    if (MODE == 0) p[r] = a[0] + a[1] + a[2] + a[3];
}

template <typename F> static float timeit(F f, int reps = 4) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

template <int K> void sweep(double* p, int64_t n, int64_t runs, size_t lds) {
    const dim3 g((unsigned)(runs / 256)), b(256);
    float t0 = timeit([&] { hipLaunchKernelGGL((k<0, K>), g, b, lds, 0, p, n, runs, 1.0000001); });
    float t1 = timeit([&] { hipLaunchKernelGGL((k<1, K>), g, b, lds, 0, p, n, runs, 1.0000001); });
    float t2 = timeit([&] { hipLaunchKernelGGL((k<2, K>), g, b, lds, 0, p, n, runs, 1.0000001); });
    float t3 = timeit([&] { hipLaunchKernelGGL((k<3, K>), g, b, lds, 0, p, n, runs, 1.0000001); });
    float t4 = timeit([&] { hipLaunchKernelGGL((k<4, K>), g, b, lds, 0, p, n, runs, 1.0000001); });
    float t5 = timeit([&] { hipLaunchKernelGGL((k<5, K>), g, b, lds, 0, p, n, runs, 1.0000001); });
    printf("runs=%lld K=%4d : compute-only %.3f | burst %.3f | spread %.3f | 8 x b128 %.3f | saddr %.3f | 15 x b32 %.3f ms\n", (long long)runs, K, t0, t1, t2, t3, t4, t5);
}

int main() {
    const int64_t n = 1000;
    for (int64_t runs : {65536ll, 262144ll}) {
        const size_t elems = (size_t)16 * n * runs;
        double* p; if (hipMalloc(&p, elems * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
        const size_t lds = runs <= 65536 ? 82 * 1024 : 55 * 1024;
        sweep<300>(p, n, runs, lds);
        sweep<600>(p, n, runs, lds);
        sweep<1050>(p, n, runs, lds);
        sweep<1500>(p, n, runs, lds);
        (void)hipFree(p);
    }
    return 0;
}
