// Does the fp32 series' store rate depend on the bytes ONE store instruction covers?  (development aid)  65 536 runs x 1000 steps
// x 15 planes of floats, layout [plane][sample][run], one workgroup per 256 runs, non-temporal buffer stores:
//   mode 0: four wavefronts, one dword per lane: 256 B per instruction (what the MC kernel's consumers issue)
//   mode 1: ONE wavefront, dwordx4 per lane: 1 KB per instruction, the workgroup's whole row of a plane (a store wavefront
//           fed through the LDS would issue these)
//   mode 2: four wavefronts, dwordx4, the 15 planes dealt round-robin to the wavefronts
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/wbench5 tools/wbench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) fill(float* p, int n, int runs, int work) {
    const uint32_t pl = (uint32_t)n * runs * 4u;
    __amdgpu_buffer_rsrc_t rs[5];
    for (int g = 0; g < 5; ++g) rs[g] = __builtin_amdgcn_make_buffer_rsrc(p + (size_t)3 * g * n * runs, 0, -1, 0x00020000);
    float v = (float)threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (MODE == 0) {
        uint32_t voff = (blockIdx.x * 256 + threadIdx.x) * 4u;
        for (int j = 0; j < n; ++j) {
            for (int w = 0; w < work; ++w) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
#pragma unroll
            for (int c = 0; c < 15; ++c) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v + c), rs[c / 3], voff, (c % 3) * pl, 2);
            voff += runs * 4u;
        }
    } else {
        uint32_t voff = blockIdx.x * 1024u + lane * 16u;
        for (int j = 0; j < n; ++j) {
            for (int w = 0; w < work; ++w) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                if (MODE == 1 || (c & 3) == wave) {
                    const uint32_t x = __float_as_uint(v + c);
                    __builtin_amdgcn_raw_buffer_store_b128(u4{x, x + 1, x + 2, x + 3}, rs[c / 3], voff, (c % 3) * pl, 2);
                }
            }
            voff += runs * 4u;
        }
    }
}
int main() {
    const int n = 1000;
    for (int runs : {65536, 262144}) {
        float* p; hipMalloc(&p, (size_t)15 * n * runs * 4);
        for (int work : {0, 200}) {
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e9, sum = 0;
                for (int rep = 0; rep < 40; ++rep) {
                    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                    hipEventRecord(a);
                    if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(runs / 256), dim3(256), 0, 0, p, n, runs, work);
                    else if (mode == 1) hipLaunchKernelGGL(fill<1>, dim3(runs / 256), dim3(64), 0, 0, p, n, runs, work);
                    else hipLaunchKernelGGL(fill<2>, dim3(runs / 256), dim3(256), 0, 0, p, n, runs, work);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b); if (rep >= 20) { sum += ms; if (ms < best) best = ms; }
                }
                const char* names[3] = {"4 waves x dword   (256 B / instr)", "1 wave  x dwordx4 (1 KB / instr) ", "4 waves x dwordx4 (1 KB / instr) "};
                printf("runs %6d  work %3d fma/step  %s : min %.3f avg %.3f ms  %.0f GB/s\n", runs, work, names[mode], best, sum / 20,
                       15.0 * n * runs * 4 / (sum / 20) / 1e6);
            }
        }
        hipFree(p);
    }
    return 0;
}
