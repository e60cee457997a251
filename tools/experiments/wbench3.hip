// Store-pattern ceiling for the fp32 MC series (development aid): one wavefront per SIMD (65 536 runs), 1000 steps, 15 planes,
// non-temporal buffer stores -- (a) one dword per lane, plane and step (layout [plane][sample][run]); (b) pairs of samples:
// one dwordx2 per lane and plane every second step, 9 planes on even and 6 on odd steps (layout [plane][sample pair][run][2]).
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/wbench3 tools/wbench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) fill(float* p, int n, int runs, int work) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pl = (uint32_t)n * runs * 4u;
    __amdgpu_buffer_rsrc_t rs[5];
    for (int g = 0; g < 5; ++g) rs[g] = __builtin_amdgcn_make_buffer_rsrc(p + (size_t)3 * g * n * runs, 0, -1, 0x00020000);
    float v = (float)r;
    if (MODE == 0) {
        uint32_t voff = r * 4u;
        for (int j = 0; j < n; ++j) {
            for (int w = 0; w < work; ++w) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
#pragma unroll
            for (int c = 0; c < 15; ++c) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v + c), rs[c / 3], voff, (c % 3) * pl, 2);
            voff += runs * 4u;
        }
    } else {
        uint32_t voff = r * 8u;
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        for (int j = 0; j < n; j += 2) {
            for (int w = 0; w < work; ++w) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
#pragma unroll
            for (int c = 0; c < 9; ++c) __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(v + c), __float_as_uint(v - c)}, rs[c / 3], voff, (c % 3) * pl, 2);
            for (int w = 0; w < work; ++w) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
#pragma unroll
            for (int c = 9; c < 15; ++c) __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(v + c), __float_as_uint(v - c)}, rs[c / 3], voff, (c % 3) * pl, 2);
            voff += runs * 8u;
        }
    }
}
int main() {
    const int n = 1000;
    for (int runs : {65536, 262144}) {
        float* p; hipMalloc(&p, (size_t)15 * n * runs * 4);
        for (int work : {0, 100, 200}) {
            for (int mode = 0; mode < 2; ++mode) {
                float best = 1e9;
                for (int rep = 0; rep < 40; ++rep) {
                    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                    hipEventRecord(a);
                    if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(runs / 256), dim3(256), 0, 0, p, n, runs, work);
                    else hipLaunchKernelGGL(fill<1>, dim3(runs / 256), dim3(256), 0, 0, p, n, runs, work);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b); if (rep >= 20 && ms < best) best = ms;
                }
                printf("runs %6d  work %3d fma/step  %s : %.3f ms  %.0f GB/s\n", runs, work, mode ? "dwordx2 pairs (9 / 6 alternating)" : "dword x15 per step               ", best,
                       15.0 * n * runs * 4 / best / 1e6);
            }
        }
        hipFree(p);
    }
    return 0;
}
