// Issue cost of the instruction kinds the fp32 / producer code is made of, gfx950: shader cycles (s_memtime ticks) per
// wave-instruction at 1..4 resident waves per SIMD, 8 independent chains (throughput) and 1 chain (dependent latency).
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/ubench3 tools/ubench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP, int CHAINS>
__global__ void k(float* out, const float* in, int iters) {
    float a[8]; uint32_t u[8]; v2f p[8]; double d[8]; uint64_t w[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[i] + threadIdx.x * 1e-6f; u[i] = (uint32_t)(in[i] * 1000) + threadIdx.x; p[i] = v2f{a[i], a[i] + 1.f}; d[i] = a[i]; w[i] = u[i]; }
    const float c = in[8], e = in[9];
    const v2f pc = {c, c}, pe = {e, e};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int i = r % CHAINS;
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pe));
            if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (OP == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (OP == 5) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(iters));
            if (OP == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"((uint32_t)w[i]), "v"(u[0]) : "vcc");
            if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (OP == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : );
            if (OP == 10) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[i]));
            if (OP == 11) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            if (OP == 12) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
            if (OP == 13) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (OP == 14) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (OP == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
            if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (OP == 18) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (OP == 19) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (OP == 20) asm volatile("s_mov_b32 s20, 0x3f811111" : : : "s20");
            if (OP == 21) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
            if (OP == 22) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + u[i] + p[i].x + p[i].y + (float)d[i] + (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[gridDim.x * blockDim.x / 2 + 1] = t1 - t0;
}
template <int OP> void run(const char* name) {
    printf("%-16s", name);
    for (int chains : {8, 1})
        for (int w : {1, 2, 4, 8}) {
            if (chains == 1 && w > 1) continue;
            const int blocks = 256 * w, tb = 256, iters = 200;
            float *out, *in; hipMalloc(&out, (blocks * tb + 64) * 4); hipMalloc(&in, 16 * 4);
            float h[16]; for (int i = 0; i < 16; ++i) h[i] = 1.0f + 0.37f * i; h[8] = 1.0000001f; h[9] = 1e-7f;
            hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (chains == 8) hipLaunchKernelGGL((k<OP, 8>), dim3(blocks), dim3(tb), 0, 0, out, in, iters * 10);
                else hipLaunchKernelGGL((k<OP, 1>), dim3(blocks), dim3(tb), 0, 0, out, in, iters * 10);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            long long cyc; hipMemcpy(&cyc, (char*)out + ((size_t)blocks * tb / 2 + 1) * 8, 8, hipMemcpyDeviceToHost);
            // wall-clock ns per wave-instruction PER SIMD (all SIMDs busy with w waves each): the issue / execution cost
            const double ns = (double)ms * 1e6 / ((double)iters * 10 * REP * w);
            if (chains == 8) printf("  w%d %5.2f (%4.1f tk)", w, ns, (double)cyc / (iters * 10 * REP));
            else printf("   | dep %5.2f ns", ns);
            hipFree(out); hipFree(in);
        }
    printf("\n");
}
int main() {
    printf("ns per wave-instruction per SIMD with w resident waves per SIMD, 8 independent chains (s_memtime ticks per instruction of one wave in brackets); dependent chain, one wave\n");
    run<0>("v_fma_f32"); run<3>("v_mul_f32"); run<1>("v_pk_fma_f32"); run<2>("v_pk_mul_f32"); run<19>("v_pk_add_f32");
    run<4>("v_add_u32"); run<5>("v_bitop3_b32"); run<17>("v_mov_b32"); run<9>("v_cndmask_b32"); run<16>("v_cmp_lt_f32");
    run<6>("v_mad_u64_u32"); run<7>("v_mul_hi_u32"); run<14>("v_mul_lo_u32"); run<12>("v_lshl_add_u64");
    run<10>("v_cvt_f32_u32"); run<11>("v_cvt_f32_f64"); run<13>("v_add_f64"); run<22>("v_fma_f64");
    run<8>("v_sqrt_f32"); run<15>("v_rcp_f32"); run<18>("v_log_f32"); run<21>("v_sin_f32"); run<20>("s_mov_b32");
    return 0;
}
