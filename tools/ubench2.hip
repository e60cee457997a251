// Second microbenchmark: compare/select/min-max/round costs with inline asm (no compiler folding).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int OP>
__global__ void k(double* out, const double* in, int iters) {
    double a = in[0] + threadIdx.x * 1e-9, b = in[1], c = in[2], d = in[3], e = in[4] + threadIdx.x, f = in[5];
    uint32_t u = threadIdx.x, v = blockIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u) : "v"(u), "v"(v) : );
            if (OP == 2) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc" : "=v"(u) : "v"(a), "v"(b), "v"(u), "v"(v) : "vcc");
            if (OP == 3) asm volatile("v_max_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
            if (OP == 4) asm volatile("v_rndne_f64 %0, %1" : "=v"(c) : "v"(a));
            if (OP == 5) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u) : "v"(a));
            if (OP == 6) asm volatile("v_cmp_gt_f64 s[20:21], %1, %2\n s_nop 1\n v_cndmask_b32 %0, %3, %4, s[20:21]" : "=v"(u) : "v"(a), "v"(b), "v"(u), "v"(v) : "s20", "s21");
            if (OP == 7) asm volatile("v_cmp_eq_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc" : "=v"(u) : "v"(u), "v"(v), "v"(u), "v"(v) : "vcc");
            if (OP == 8) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
            if (OP == 9) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u) : "v"(u), "v"(v));
            if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(e) : "v"(u), "v"(v), "v"(e) : "vcc");
            if (OP == 11) asm volatile("s_mov_b32 s20, 0x3f811111" ::: "s20");
            if (OP == 12) asm volatile("s_add_u32 s20, s20, 17" ::: "s20", "scc");
            if (OP == 13) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(u) : "s20");
            if (OP == 14) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
            if (OP == 15) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
            if (OP == 16) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(c) : "v"(u));
            if (OP == 17) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(u) : "v"(u));
            if (OP == 18) asm volatile("v_add_u32 %0, %1, %2" : "=v"(u) : "v"(u), "v"(v));
            if (OP == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(v));
            if (OP == 20) asm volatile("v_fma_f64 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(a), "v"(b) : "s20","s21");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + e + u;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[gridDim.x * blockDim.x] = t1 - t0;
}
template <int OP> void run(const char* name) {
    for (int w : {1, 2}) {
        const int blocks = 256 * w, tb = 256, iters = 400;
        double *out, *in; (void)hipMalloc(&out, (blocks * tb + 8) * 8); (void)hipMalloc(&in, 16 * 8);
        double h[16]; for (int i = 0; i < 16; ++i) h[i] = 1.0 + 0.37 * i;
        (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(tb), 0, 0, out, in, 10);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(tb), 0, 0, out, in, iters);
        (void)hipDeviceSynchronize();
        long long cyc; (void)hipMemcpy(&cyc, (char*)out + (size_t)blocks * tb * 8, 8, hipMemcpyDeviceToHost);
        printf("%-40s waves/SIMD=%d : %6.2f ticks/op\n", name, w, (double)cyc / (iters * REP));
        (void)hipFree(out); (void)hipFree(in);
    }
}
int main() {
    run<8>("v_fma_f64 (dependent)"); run<20>("v_fma_f64 with SGPR operand"); run<14>("v_mul_f64"); run<15>("v_add_f64");
    run<3>("v_max_f64"); run<4>("v_rndne_f64"); run<5>("v_cvt_i32_f64"); run<16>("v_cvt_f64_u32");
    run<0>("v_cmp_gt_f64 -> vcc"); run<1>("v_cndmask_b32 (vcc)"); run<2>("v_cmp_gt_f64 + v_cndmask (vcc)");
    run<6>("v_cmp_gt_f64 -> sgpr, nop, cndmask"); run<7>("v_cmp_eq_u32 + v_cndmask");
    run<9>("v_xor_b32"); run<18>("v_add_u32"); run<17>("v_lshlrev_b32"); run<19>("v_mov_b32"); run<10>("v_mad_u64_u32");
    run<11>("s_mov_b32 literal"); run<12>("s_add_u32"); run<13>("v_readlane_b32");
    return 0;
}
