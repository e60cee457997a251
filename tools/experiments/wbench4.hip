// Cache-policy sweep for the MC store pattern (development aid): 65 536 / 262 144 runs x 1000 steps x 15 planes, one wavefront
// per 64 runs, buffer stores with aux = 0 (default), 1 (sc0), 2 (nt), 3 (sc0 nt), 16 (sc1), 17, 18 (sc1 nt), 19; fp32 (dword)
// and fp64 (dwordx2) element size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <int AUX, int ES>
__global__ void __launch_bounds__(256) fill(char* p, int n, int runs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pl = (uint32_t)n * runs * ES;
    __amdgpu_buffer_rsrc_t rs[15];
    for (int g = 0; g < 15; ++g) rs[g] = __builtin_amdgcn_make_buffer_rsrc(p + (size_t)g * n * runs * ES, 0, -1, 0x00020000);
    float v = (float)r;
    uint32_t voff = r * ES;
    for (int j = 0; j < n; ++j) {
        v = __builtin_fmaf(v, 1.0000001f, 0.5f);
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            if (ES == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v + c), rs[c], voff, 0, AUX);
            else __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(v + c), __float_as_uint(v - c)}, rs[c], voff, 0, AUX);
        }
        voff += runs * ES;
    }
}
template <int AUX, int ES> float run(char* p, int n, int runs) {
    float best = 1e9;
    for (int rep = 0; rep < 30; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL((fill<AUX, ES>), dim3(runs / 256), dim3(256), 0, 0, p, n, runs);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (rep >= 15 && ms < best) best = ms;
    }
    return best;
}
template <int ES> void sweep(char* p, int n, int runs) {
    const double gb = 15.0 * n * runs * ES / 1e9;
    float t[8] = {run<0, ES>(p, n, runs), run<1, ES>(p, n, runs), run<2, ES>(p, n, runs), run<3, ES>(p, n, runs),
                  run<16, ES>(p, n, runs), run<17, ES>(p, n, runs), run<18, ES>(p, n, runs), run<19, ES>(p, n, runs)};
    const int aux[8] = {0, 1, 2, 3, 16, 17, 18, 19};
    printf("runs %6d  %d B/lane  %.2f GB:", runs, ES, gb);
    for (int i = 0; i < 8; ++i) printf("  aux%-2d %.3f ms (%.0f)", aux[i], t[i], gb / t[i] * 1e3);
    printf("\n");
}
int main() {
    const int n = 1000;
    for (int runs : {65536, 262144}) {
        char* p; hipMalloc(&p, (size_t)15 * n * runs * 8);
        sweep<4>(p, n, runs);
        sweep<8>(p, n, runs);
        hipFree(p);
    }
    return 0;
}
