// Elementary functions of the MC kernel, specialised to the argument ranges the kernel needs.
//
//   * the Box-Muller transform of the normal generator, DEFINED operation by operation in IEEE single precision (every
//     product, sum and square root rounded separately -- no fused multiply-adds --, two committed lookup tables,
//     normal_tables.inc), so that the NumPy and the C oracle (oracle/) reproduce the device's normals to the
//     bit (tests/test_gpu_parity.py::test_rng_words_bit_exact_and_normals, ::test_box_muller_corner_cases).  Single
//     precision because the SIMD issues an fp32 instruction in half the time of an fp64 one and the twelve normals of a
//     step were 40 % of the fused kernel's fp64 work (round 2, third definition of the stream: 1.60 -> 1.46 ms per launch).
//   * sin/cos(a + d) from sin/cos(a) for small |d| in fp64 -> Euler-angle attitude propagation (a few ulp; the engine's
//     parity tolerances are 1e-12..1e-9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ginsim {

#define GINSIM_FM __device__ __forceinline__

// A loop-invariant fp64 constant held in a VGPR pair.  v_fma_f64 cannot take a 64-bit literal, so every
// polynomial coefficient lives in registers: left to the compiler they become SGPR pairs that are either hoisted
// out of the time loop and spilled to VGPR lanes (v_readlane per use) or re-materialised with two s_mov_b32 per
// use (~220 scalar moves per step, which a lone wavefront on its SIMD cannot hide).  The empty asm makes the
// value opaque, so it is materialised once and stays in vector registers.
GINSIM_FM double vconst(double k) {
    int lo = __double2loint(k), hi = __double2hiint(k);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return __hiloint2double(hi, lo);
}

struct MathConsts {
    double sc[5];           // sin: -1/3!, 1/5!, -1/7!, 1/9!, -1/11!
    double cc[6];           // cos: -1/2!, 1/4!, ... 1/12!
    // OPAQUE = true pins the 11 constants in VGPRs (22 registers); false leaves them to the compiler (SGPR literals),
    // which is what the two-algorithm kernels need to stay under 256 VGPRs without scratch spills.
    template <bool OPAQUE>
    GINSIM_FM void init() {
        auto vconst = [](double x) { return OPAQUE ? ginsim::vconst(x) : x; };
        const double s[5] = {-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0};
        const double c[6] = {-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0, 1.0 / 479001600.0};
#pragma unroll
        for (int k = 0; k < 5; ++k) sc[k] = vconst(s[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) cc[k] = vconst(c[k]);
    }
};

// Box-Muller lookup tables (fp32, 8 KB), copied into LDS by every workgroup from the committed constants:
//   lg[k] = {c_k, -2/c_k, -2 ln c_k, 0}, c_k the centre of the k-th of 256 bins of the mantissa m in [sqrt(1/2), sqrt(2))
//           (bins of 2^15 consecutive bit patterns from 0x3f3504f3; c = 1 and ln c = 0 for the bin that contains 1, so that
//           ln u -> 0 without cancellation as u -> 1); the factor -2 of the radius sqrt(-2 ln u) is folded in;
//   sc[i] = {sin a_i, cos a_i}, a_i = 2 pi (i + 1/2) / 512: the centre of the i-th of 512 sectors of the turn.
constexpr int kLogBins = 256, kAngBins = 512;
constexpr int kNormalTableWords = 4 * kLogBins + 2 * kAngBins;
static __device__ const uint32_t kNormalTableBits[kNormalTableWords] = {
#include "normal_tables.inc"
};
struct NormalTables {
    const float4* lg;
    const float2* sc;
};

GINSIM_FM NormalTables fill_normal_tables(uint32_t* lds, int tid, int nthreads) {
    for (int k = tid; k < kNormalTableWords; k += nthreads) lds[k] = kNormalTableBits[k];
    return NormalTables{reinterpret_cast<const float4*>(lds), reinterpret_cast<const float2*>(lds + 4 * kLogBins)};
}

// x = -2 ln u for u = (f32(a) + 1/2) 2^-32 in (0, 1], a = a 32-bit word: the squared Box-Muller radius (|z| <= 6.8).
//   u = m 2^e, m in [sqrt(1/2), sqrt(2));  x = e (-2 ln 2) + (-2 ln c_k) + (r + r^2 (1/4 + r/12)),  r = (m - c_k)(-2/c_k)
// m - c_k is exact (|m - c_k| <= 2^-8), |r| <= 2^-7, the series is cut below 2^-26 r.  The exponent/mantissa split and the
// bin index are integer arithmetic on the bit pattern: adding (0x3f800000 - 0x3f3504f3) moves the sqrt(1/2) boundary
// onto an exponent boundary.  EVERY operation below is one IEEE single-precision operation, in this order: the oracles
// repeat them (oracle/philox.py radius2_f32).
GINSIM_FM float radius2_f32(uint32_t a, const NormalTables& tab) {
#pragma clang fp contract(off)
    const float t = (float)a;
    const float u = (t + 0.5f) * 0x1.0p-32f;
    const uint32_t hx = __float_as_uint(u) + (0x3f800000u - 0x3f3504f3u);
    const float ef = (float)((int)(hx >> 23) - 127);
    const float4 k = tab.lg[(hx >> 15) & (kLogBins - 1)];
    const float m = __uint_as_float((hx & 0x007fffffu) + 0x3f3504f3u);
    const float d = m - k.x;
    const float r = d * k.y;
    float q = r * (1.0f / 12.0f);
    q = q + 0.25f;
    const float r2 = r * r;
    q = q * r2;
    const float small = r + q;
    float x = ef * -1.3862943611198906f;
    x = x + k.z;
    x = x + small;
    return x;
}

// The same on TWO words at once: the arithmetic as packed single-precision operations (v_pk_mul_f32 / v_pk_add_f32 do two
// IEEE operations per lane in the issue time of one), the bit manipulation and the table reads per element.  Bit for bit
// the scalar function on each element.
typedef float v2f __attribute__((ext_vector_type(2)));
GINSIM_FM v2f radius2_f32x2(uint32_t a0, uint32_t a1, const NormalTables& tab) {
#pragma clang fp contract(off)
    const v2f t = {(float)a0, (float)a1};
    const v2f u = (t + 0.5f) * 0x1.0p-32f;
    const uint32_t h0 = __float_as_uint(u.x) + (0x3f800000u - 0x3f3504f3u), h1 = __float_as_uint(u.y) + (0x3f800000u - 0x3f3504f3u);
    const v2f ef = {(float)((int)(h0 >> 23) - 127), (float)((int)(h1 >> 23) - 127)};
    const float4 k0 = tab.lg[(h0 >> 15) & (kLogBins - 1)], k1 = tab.lg[(h1 >> 15) & (kLogBins - 1)];
    const v2f m = {__uint_as_float((h0 & 0x007fffffu) + 0x3f3504f3u), __uint_as_float((h1 & 0x007fffffu) + 0x3f3504f3u)};
    const v2f kc = {k0.x, k1.x}, ki = {k0.y, k1.y}, kl = {k0.z, k1.z};
    const v2f d = m - kc;
    const v2f r = d * ki;
    v2f q = r * (1.0f / 12.0f);
    q = q + 0.25f;
    const v2f r2 = r * r;
    q = q * r2;
    const v2f small = r + q;
    v2f x = ef * -1.3862943611198906f;
    x = x + kl;
    x = x + small;
    return x;
}

GINSIM_FM void sincos_f32x2(uint32_t w0, uint32_t w1, v2f& sn, v2f& cs, const NormalTables& tab) {
#pragma clang fp contract(off)
    const float2 t0 = tab.sc[(w0 >> 15) & (kAngBins - 1)], t1 = tab.sc[(w1 >> 15) & (kAngBins - 1)];
    const v2f ts = {t0.x, t1.x}, tc = {t0.y, t1.y};
    v2f b = {(float)(int)(w0 & 0x7fffu), (float)(int)(w1 & 0x7fffu)};
    b = b + (0.5f - 16384.0f);
    b = b * 3.7450703562e-07f;
    const v2f tt = b * b;
    v2f u1 = tt * (-1.0f / 6.0f);
    u1 = u1 * b;
    const v2f sb = b + u1;
    const v2f cm = tt * -0.5f;
    v2f p1 = tc * sb;
    const v2f p2 = ts * cm;
    p1 = p1 + p2;
    sn = ts + p1;
    v2f q1 = tc * cm;
    const v2f q2 = ts * sb;
    q1 = q1 - q2;
    cs = tc + q1;
}

// Correctly rounded sqrt(x) for 0 <= x < 2^7 (never denormal here: x = 0 or x >= 2^-24): v_sqrt_f32 is good to 1 ulp;
// of its result and the two neighbours the one whose square brackets x is the rounded root (the residuals are exact in
// one fma each).  The oracles call sqrtf / np.sqrt, which are correctly rounded.
GINSIM_FM float sqrt_rn_f32(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vdn = __builtin_fmaf(-dn, s, x), vup = __builtin_fmaf(-up, s, x);
    s = vdn <= 0.0f ? dn : s;
    s = vup > 0.0f ? up : s;
    return s;
}

// sin and cos of the Box-Muller angle 2 pi (w24 + 1/2) 2^-24, w24 = the low 24 bits of a Philox word: sector i = top 9
// bits, b = centred remainder in radians (|b| <= pi/512), angle = a_i + b;  sin b = b - b^3/6 (next term 7e-14),
// cos b - 1 = -b^2/2 (next term 6e-11).  One IEEE single-precision operation per line (oracle/philox.py sincos_f32).
GINSIM_FM void sincos_f32(uint32_t w, float& sn, float& cs, const NormalTables& tab) {
#pragma clang fp contract(off)
    const float2 t = tab.sc[(w >> 15) & (kAngBins - 1)];
    float b = (float)(int)(w & 0x7fffu);
    b = b + (0.5f - 16384.0f);
    b = b * 3.7450703562e-07f;                  // f32(2 pi 2^-24)
    const float tt = b * b;
    float u1 = tt * (-1.0f / 6.0f);
    u1 = u1 * b;
    const float sb = b + u1;
    const float cm = tt * -0.5f;
    float p1 = t.y * sb;
    const float p2 = t.x * cm;
    p1 = p1 + p2;
    sn = t.x + p1;
    float q1 = t.y * cm;
    const float q2 = t.x * sb;
    q1 = q1 - q2;
    cs = t.y + q1;
}

// 1/x to ~1 ulp: hardware v_rcp_f64 estimate + two Newton steps.
GINSIM_FM double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// 1/x to ~2^-46 (one Newton step on the 2^-23 hardware estimate): enough where a residual correction of the quotient
// follows (q = a y; q += (a - x q) y is then ~1 ulp).
GINSIM_FM double rcp_n1(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}

// 1/sqrt(x) to ~1 ulp for normal x: v_rsq_f64 estimate + two Newton steps y += y (1/2 - x y^2 / 2).
GINSIM_FM double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(y, __builtin_fma(-0.5 * y, x * y, 0.5), y);
    return __builtin_fma(y, __builtin_fma(-0.5 * y, x * y, 0.5), y);
}

// Rotate (s,c) = (sin a, cos a) by a small angle d: returns sin/cos(a+d).  Valid for |d| <= 0.25 rad
// (truncation < 3e-18); callers fall back to an exact sincos for larger steps.
GINSIM_FM void rotate_sincos(double d, double& s, double& c) {
    const double t = d * d;
    double ps = -1.0 / 39916800.0;                              // -1/11!
    ps = __builtin_fma(ps, t, 1.0 / 362880.0);
    ps = __builtin_fma(ps, t, -1.0 / 5040.0);
    ps = __builtin_fma(ps, t, 1.0 / 120.0);
    ps = __builtin_fma(ps, t, -1.0 / 6.0);
    const double sd = __builtin_fma(d * t, ps, d);              // sin d
    double pc = 1.0 / 479001600.0;                              // 1/12!
    pc = __builtin_fma(pc, t, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, t, 1.0 / 40320.0);
    pc = __builtin_fma(pc, t, -1.0 / 720.0);
    pc = __builtin_fma(pc, t, 1.0 / 24.0);
    pc = __builtin_fma(pc, t, -0.5);
    const double cm1 = t * pc;                                  // cos d - 1
    const double s0 = s, c0 = c;
    s = __builtin_fma(c0, sd, __builtin_fma(s0, cm1, s0));      // s + s (cos d - 1) + c sin d
    c = __builtin_fma(-s0, sd, __builtin_fma(c0, cm1, c0));     // c + c (cos d - 1) - s sin d
}

// ---- the same rotation with its coefficients taken from VGPR-resident constants (hot loop) -------------
GINSIM_FM void rotate_sincos(double d, double& s, double& c, const MathConsts& k) {
    const double t = d * d;
    double ps = k.sc[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) ps = __builtin_fma(ps, t, k.sc[i]);
    const double sd = __builtin_fma(d * t, ps, d);
    double pc = k.cc[5];
#pragma unroll
    for (int i = 4; i >= 0; --i) pc = __builtin_fma(pc, t, k.cc[i]);
    const double cm1 = t * pc;
    const double s0 = s, c0 = c;
    s = __builtin_fma(c0, sd, __builtin_fma(s0, cm1, s0));
    c = __builtin_fma(-s0, sd, __builtin_fma(c0, cm1, c0));
}

// The same rotation for |d| <= 2^-6 rad (a step of 0.9 deg: every sample of a vehicle profile at >= 100 Hz): three
// series terms each instead of five / six (next terms d^9/9! and d^8/8!: < 1e-19 relative).
GINSIM_FM void rotate_sincos_small(double d, double& s, double& c, const MathConsts& k) {
    const double t = d * d;
    const double ps = __builtin_fma(__builtin_fma(k.sc[2], t, k.sc[1]), t, k.sc[0]);
    const double sd = __builtin_fma(d * t, ps, d);
    const double pc = __builtin_fma(__builtin_fma(k.cc[2], t, k.cc[1]), t, k.cc[0]);
    const double cm1 = t * pc;
    const double s0 = s, c0 = c;
    s = __builtin_fma(c0, sd, __builtin_fma(s0, cm1, s0));
    c = __builtin_fma(-s0, sd, __builtin_fma(c0, cm1, c0));
}

}  // namespace ginsim
