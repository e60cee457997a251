// Pure-write bandwidth of MI355X for the MC kernel's store pattern (development aid): what would a kernel that
// only writes the [15][n][runs] planes achieve?  Gives the empirical ceiling the materialising kernel is held to.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void __launch_bounds__(256) fill_linear(double* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (double)i;
}

// one lane = one run, time loop inside, PLANES planes of [n][runs]; UNROLL steps between dependent updates
template <typename T, int PLANES>
__global__ void __launch_bounds__(256) fill_mc(T* p, int64_t n, int64_t runs) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= runs) return;
    const int64_t plane = n * runs;
    T v = (T)r;
    for (int64_t j = 0; j < n; ++j) {
        v = v * (T)1.0000001 + (T)0.5;
#pragma unroll
        for (int c = 0; c < PLANES; ++c) p[c * plane + j * runs + r] = v + (T)c;
    }
}

template <typename F> static float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int64_t n = 1000;
    for (int64_t runs : {65536ll, 131072ll, 262144ll}) {
        const size_t elems = (size_t)15 * n * runs;
        double* p; if (hipMalloc(&p, elems * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
        const double gb = elems * 8 / 1e9;
        float ms = timeit([&] { hipMemsetAsync(p, 0, elems * 8, 0); });
        printf("runs=%lld  %.2f GB  hipMemsetAsync      : %.3f ms  %.0f GB/s\n", (long long)runs, gb, ms, gb / ms * 1e3);
        for (int blocks : {1024, 4096, 16384}) {
            ms = timeit([&] { hipLaunchKernelGGL(fill_linear, dim3(blocks), dim3(256), 0, 0, p, elems); });
            printf("runs=%lld  fill_linear %5d blocks        : %.3f ms  %.0f GB/s\n", (long long)runs, blocks, ms, gb / ms * 1e3);
        }
        ms = timeit([&] { hipLaunchKernelGGL((fill_mc<double, 15>), dim3((unsigned)(runs / 256)), dim3(256), 0, 0, p, n, runs); });
        printf("runs=%lld  fill_mc<double,15> (MC pattern)   : %.3f ms  %.0f GB/s\n", (long long)runs, ms, gb / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((fill_mc<float, 15>), dim3((unsigned)(runs / 256)), dim3(256), 0, 0, (float*)p, n, runs); });
        printf("runs=%lld  fill_mc<float,15>  (fp32 pattern) : %.3f ms  %.0f GB/s\n", (long long)runs, ms, gb / 2 / ms * 1e3);
        hipFree(p);
    }
    return 0;
}
