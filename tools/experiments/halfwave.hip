// Does a wavefront whose upper 32 lanes are switched off issue a VALU instruction in half the time on gfx950 (SIMD-32)?
//   hipcc --offload-arch=gfx950 -O3 -o halfwave tools/halfwave.hip && ./halfwave
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k(double* out, int iters, int active) {
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 0.999999, c = 1e-9, d = a + 1.0, e = a + 2.0, f = a + 3.0;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c);
            a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c);
        }
        out[blockIdx.x * 64 + lane] = a + d + e + f;
    }
}
__global__ void __launch_bounds__(64) k32(float* out, int iters, int active) {
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-6f, b = 0.9999f, c = 1e-6f, d = a + 1.0f, e = a + 2.0f, f = a + 3.0f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
            a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
        }
        out[blockIdx.x * 64 + lane] = a + d + e + f;
    }
}
int main() {
    double* out; hipMalloc(&out, 8 * 64 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
        for (int active : {64, 32, 16}) {
            const int blocks = 1024 * waves_per_simd, iters = 20000;
            for (int p = 0; p < 2; ++p) {
                hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 200, active);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (p == 0) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, active);
                else hipLaunchKernelGGL(k32, dim3(blocks), dim3(64), 0, 0, (float*)out, iters, active);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%s waves/SIMD %d active lanes %2d: %.3f ms -> %.2f cycles per FMA per wave at 2.4 GHz\n", p ? "f32" : "f64", waves_per_simd, active, ms,
                       ms * 1e-3 * 2.4e9 / (8.0 * iters) / waves_per_simd);
            }
        }
    return 0;
}
