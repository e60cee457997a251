// Elementary functions of the MC kernel, specialised to the argument ranges the kernel needs.
//
//   * the normal transform of the generator (normal_icdf: piecewise-cubic inversion of the tail probability, one word per
//     normal), DEFINED operation by operation in IEEE single precision on a committed coefficient table
//     (normal_tables.inc), so that the NumPy and the C oracle (oracle/) reproduce the device's normals to the bit
//     (tests/test_gpu_parity.py::test_rng_words_bit_exact_and_normals, ::test_normal_transform_corner_cases).
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

// The normal generator's coefficient table (fp32, 3968 B of committed constants), copied into LDS by every workgroup:
//   q[seg] = {c0, c1, c2, c3}, seg = (lz - 1) 8 + sub: the cubic that inverts the upper tail probability on the sub-th
//            eighth of the octave [2^-(lz+1), 2^-lz) of t (tools/gen_normal_tables.py says how it was fitted).
// In LDS the eight cubics of octave lz sit in the UPPER half of the 256-byte block lz (8 KB in all, half of it unused), so
// that the byte offset of a segment is the top of the 64-bit value {lz, y} -- (lz << 8) | (y >> 24), one v_alignbit_b32 --
// with its low four bits cleared: bit 7 of y >> 24 is the leading one of the shifted magnitude (the upper half), bits 6..4
// are the eighth.  Two instructions per normal instead of three (shift, shift, and-or) with the dense table of round 2.
constexpr int kNormalOctaves = 31, kNormalSubs = 8;
constexpr int kNormalTableWords = 4 * kNormalOctaves * kNormalSubs;         // the committed constants
constexpr int kNormalLdsWords = 64 * (kNormalOctaves + 1);                   // their copy in LDS: blocks 0 .. 31 of 256 bytes
static __device__ const uint32_t kNormalTableBits[kNormalTableWords] = {
#include "normal_tables.inc"
};
struct NormalTables {
    const char* base;
};

GINSIM_FM NormalTables fill_normal_tables(uint32_t* lds, int tid, int nthreads) {
    for (int k = tid; k < kNormalTableWords; k += nthreads) lds[((k >> 5) + 1) * 64 + 32 + (k & 31)] = kNormalTableBits[k];
    return NormalTables{reinterpret_cast<const char*>(lds)};
}

// (z & 0x7fffffff) | (w & 0x80000000) in one v_bfi_b32 (the compiler emits v_and_b32 + v_and_or_b32 for the expression)
GINSIM_FM float with_sign_of(float z, uint32_t w) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0x7fffffffu), "v"(z), "v"(w));
    return __uint_as_float(r);
}

// One standard normal from ONE 32-bit word by inversion, DEFINED operation by operation (the oracles repeat it to the bit:
// oracle/philox.py normal_icdf):
//   m = (w & 0x7fffffff) | 1     the magnitude, odd: upper tail probability t = m 2^-32 in (0, 1/2); |z| <= 6.23 (m = 1)
//   lz = clz(m), y = m << lz     octave of t and, from the bits below the leading one, the eighth of the octave (3 bits)
//   x                            the next 23 bits as a float in [1, 2) (v_alignbit_b32 drops them under the exponent of 1.0f)
//   z = c0 + x (c1 + x (c2 + x c3))   three fused multiply-adds on the segment's coefficients (one 16-byte LDS read)
//   sign of z = bit 31 of w
// Integer operations, one table read and three fmas per normal -- no logarithm, square root or sin / cos.  Round 2's
// single-precision Box-Muller needed ~30 ns of SIMD time per normal (packed arithmetic, the correctly rounded square root,
// two table reads per pair), this ~13: the fused fp64 kernel went from 1.39 to 1.23 ms per C2 launch with it.
GINSIM_FM float normal_icdf(uint32_t w, const NormalTables& tab) {
    const uint32_t m = (w & 0x7fffffffu) | 1u;
    const int lz = __builtin_clz(m);                                    // 1 .. 31
    const uint32_t y = m << lz;
    const float4 c = *reinterpret_cast<const float4*>(tab.base + (__builtin_amdgcn_alignbit((uint32_t)lz, y, 24) & ~15u));
    const float x = __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, y << 4, 9));      // 0x3f800000 | ((y << 4) >> 9)
    const float z = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c.w, x, c.z), x, c.y), x, c.x);
    return with_sign_of(z, w);
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
