// Feasibility of a RUN-MAJOR streaming Allan level pass (development aid): lane = series, rows = time, nine bin machines driven by
// wave-uniform counters over one prefix sum, D rows of prefetch.  Times level 0 of config 5's shape: S = 192 series x n = 1 440 000.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/allan_rm_bench tools/allan_rm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Lv { int64_t n, n_out, lim[9]; int64_t L; int S; };

__global__ void fill(double* x, int64_t count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    x[i] = -9.79 + 5e-3 * ((double)(h & 0xFFFFF) / 524288.0 - 1.0);
}

template <int D, bool MACHINES>
__global__ void __launch_bounds__(64) allan_rm(const double* __restrict__ x, double* __restrict__ out, double* __restrict__ partial, const Lv lv) {
    const int lane = threadIdx.x;
    const int sigma = blockIdx.y * 64 + lane;
    const bool active = sigma < lv.S;
    const int64_t T0 = (int64_t)blockIdx.x * lv.L;
    const int64_t T1 = T0 + lv.L < lv.n ? T0 + lv.L : lv.n;
    const int64_t Tend = T1 + 20 < lv.n ? T1 + 20 : lv.n;
    const double* col = x + (active ? sigma : 0);
    const int64_t S = lv.S;
    const double x0 = col[T0 * S];
    double P = 0.0, Q[10], Sp[10], acc[9];
    int64_t nxt[10];
    bool started[10], have[10];
#pragma unroll
    for (int j = 1; j <= 10; ++j) {
        const int64_t b0 = (T0 + j - 1) / j * j;
        started[j - 1] = b0 == T0;
        nxt[j - 1] = started[j - 1] ? T0 + j : b0;
        Q[j - 1] = 0.0; Sp[j - 1] = 0.0; have[j - 1] = false;
        if (j <= 9) acc[j - 1] = 0.0;
    }
    int cnt[10];                                // entries until machine j's next event (wave-uniform)
#pragma unroll
    for (int j = 1; j <= 10; ++j) cnt[j - 1] = (int)(nxt[j - 1] - T0);
    auto load = [&](double (&buf)[D], int64_t tb) {
#pragma unroll
        for (int i = 0; i < D; ++i) { const int64_t t = tb + i; buf[i] = col[(t < lv.n ? t : lv.n - 1) * S]; }
    };
    auto process = [&](const double (&buf)[D], int64_t tb) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int64_t t1 = tb + i + 1;              // entries consumed once this one is in
            if (t1 <= Tend) {
                P += buf[i] - x0;
#pragma unroll
                for (int j = 1; j <= 10; ++j) {
                    if (MACHINES && --cnt[j - 1] == 0) {            // wave-uniform
                        cnt[j - 1] = j;
                        if (!started[j - 1]) { started[j - 1] = true; Q[j - 1] = P; }
                        else {
                            const double Sb = P - Q[j - 1];
                            Q[j - 1] = P;
                            const bool own = t1 - 2 * j < T1;
                            if (j <= 9) {
                                if (have[j - 1] && own && t1 <= lv.lim[j - 1]) { const double d = Sb - Sp[j - 1]; acc[j - 1] = __builtin_fma(d, d, acc[j - 1]); }
                            } else if (t1 - 10 < T1 && t1 - 10 >= T0 && t1 / 10 - 1 < lv.n_out && active) {
                                out[(t1 / 10 - 1) * S + sigma] = __builtin_fma(10.0, x0, Sb);
                            }
                            Sp[j - 1] = Sb; have[j - 1] = true;
                        }
                    }
                }
            }
        }
    };
    double bufA[D], bufB[D];
    load(bufA, T0);
    for (int64_t tb = T0; tb < Tend; tb += 2 * D) {
        load(bufB, tb + D);
        process(bufA, tb);
        load(bufA, tb + 2 * D);
        process(bufB, tb + D);
    }
    if (active)
#pragma unroll
        for (int j = 0; j < 9; ++j) partial[((int64_t)blockIdx.x * 9 + j) * S + sigma] = acc[j] + P;
}

int main() {
    const int S = 192; const int64_t n = 1440000;
    double *x, *out, *partial;
    hipMalloc(&x, sizeof(double) * n * S); hipMalloc(&out, sizeof(double) * (n / 10 + 1) * S);
    hipLaunchKernelGGL(fill, dim3((unsigned)((n * S + 255) / 256)), dim3(256), 0, 0, x, n * S);
    for (int64_t L : {525, 1050, 2100, 4200}) {
        Lv lv; lv.n = n; lv.n_out = n / 10; lv.L = L; lv.S = S;
        for (int j = 1; j <= 9; ++j) lv.lim[j - 1] = n / j * j;
        const int ranges = (int)((n + L - 1) / L);
        hipMalloc(&partial, sizeof(double) * ranges * 9 * S);
        for (int d : {16, 32}) {
            float best = 1e9, sum = 0;
            for (int rep = 0; rep < 30; ++rep) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (d == 16) hipLaunchKernelGGL((allan_rm<16, true>), dim3(ranges, (S + 63) / 64), dim3(64), 0, 0, x, out, partial, lv);
                else hipLaunchKernelGGL((allan_rm<16, false>), dim3(ranges, (S + 63) / 64), dim3(64), 0, 0, x, out, partial, lv);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (rep >= 10) { sum += ms; if (ms < best) best = ms; }
            }
            printf("L %5lld  mode %2d (16: machines, 32: loads + prefix only)  waves %6d : min %.3f avg %.3f ms  %.0f GB/s\n", (long long)L, d, ranges * 3, best, sum / 20, 8.0 * n * S / (sum / 20) / 1e6);
        }
        hipFree(partial);
    }
    return 0;
}
