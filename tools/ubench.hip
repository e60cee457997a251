// Instruction issue-cost microbenchmark on gfx950 (development aid): cycles per wave-instruction for the
// op classes the MC kernel uses, at 1 or 2 waves per SIMD, with 8 independent chains (throughput) or 1 (latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP 64
template <int OP, int CHAINS>
__global__ void k(double* out, const double* in, int iters) {
    double a[8]; uint32_t u[8]; uint64_t w[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[i] + threadIdx.x * 1e-9; u[i] = (uint32_t)(in[i] * 1000) + threadIdx.x; w[i] = u[i]; }
    const double c = in[8], d = in[9];
    uint32_t sacc = (uint32_t)iters;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int i = r % CHAINS;
            if (OP == 0) a[i] = __builtin_fma(a[i], c, d);
            if (OP == 1) a[i] = a[i] + c;
            if (OP == 2) a[i] = a[i] * c;
            if (OP == 3) w[i] = (uint64_t)(uint32_t)w[i] * 0xD2511F53u + (w[i] >> 32);
            if (OP == 4) u[i] = u[i] * 0xCD9E8D57u;
            if (OP == 5) u[i] = u[i] ^ (u[i] >> 3);            // 2 ops
            if (OP == 6) u[i] = u[i] + 0x9E3779B9u;
            if (OP == 7) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 8) a[i] = __builtin_amdgcn_rsq(a[i]);
            if (OP == 9) a[i] = (double)u[i] + a[i];           // cvt + add
            if (OP == 10) a[i] = a[i] > c ? d : a[i];          // cmp + 2 cndmask
            if (OP == 11) { asm volatile("s_add_u32 %0, %0, 12345" : "+s"(sacc)); }
            if (OP == 12) { asm volatile("s_mov_b32 %0, 0x3f811111" : "=s"(sacc)); }
            if (OP == 13) u[i] = __builtin_amdgcn_readlane(u[i], 3) + u[i];   // readlane + add
            if (OP == 14) u[i] = __umulhi(u[i], 0xD2511F53u);
            if (OP == 15) a[i] = __builtin_amdgcn_sqrt(a[i]);
            if (OP == 16) a[i] = __builtin_amdgcn_ldexp(a[i], 1);
            if (OP == 17) a[i] = __builtin_amdgcn_trig_preop(a[i], 1);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + u[i] + (double)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[gridDim.x * blockDim.x] = t1 - t0;
}

template <int OP> void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, tb = 256, iters = 200;
    double *out, *in; hipMalloc(&out, (blocks * tb + 8) * 8); hipMalloc(&in, 16 * 8);
    double h[16]; for (int i = 0; i < 16; ++i) h[i] = 1.0 + 0.37 * i; h[8] = 1.0000001; h[9] = 1e-7;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int chains : {8, 1}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (chains == 8) { hipLaunchKernelGGL((k<OP, 8>), dim3(blocks), dim3(tb), 0, 0, out, in, 10); }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (chains == 8) hipLaunchKernelGGL((k<OP, 8>), dim3(blocks), dim3(tb), 0, 0, out, in, iters);
        else hipLaunchKernelGGL((k<OP, 1>), dim3(blocks), dim3(tb), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, (char*)out + (size_t)blocks * tb * 8, 8, hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD=%d chains=%d : %7.2f memtime-ticks/op (%.3f ms)\n", name, waves_per_simd, chains,
               (double)cyc / (iters * REP), ms);
    }
    hipFree(out); hipFree(in);
}

int main() {
    for (int w : {1, 2}) {
        run<0>("v_fma_f64", w); run<1>("v_add_f64", w); run<2>("v_mul_f64", w);
        run<3>("v_mad_u64_u32 (+shift)", w); run<4>("v_mul_lo_u32", w); run<14>("v_mul_hi_u32", w);
        run<5>("v_xor+v_lshr (2 int ops)", w); run<6>("v_add_u32", w);
        run<7>("v_rcp_f64", w); run<8>("v_rsq_f64", w); run<15>("v_sqrt_f64", w); run<16>("v_ldexp_f64", w);
        run<9>("v_cvt_f64_u32 + v_add_f64", w); run<10>("v_cmp_f64 + 2 cndmask", w);
        run<11>("s_add_u32", w); run<12>("s_mov_b32", w); run<13>("v_readlane + v_add_u32", w);
    }
    return 0;
}
