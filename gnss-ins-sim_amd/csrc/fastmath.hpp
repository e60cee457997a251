// Lean fp64 elementary functions for the MC kernel, specialised to the argument ranges the kernel needs.
//
// ROCm's OCML double-precision log / sincos / sincospi are <1 ulp and pay for it with double-double
// arithmetic and a Payne-Hanek large-argument path (~1000 v_add_f64 in the first build of mc_kernel).
// The kernel only needs:
//   * log(u) for a uniform u in (0,1]               -> Box-Muller radius
//   * sin/cos(pi*x) for x = 2u in (0,2)             -> Box-Muller angle
//   * sin/cos(a + d) from sin/cos(a) for small |d|  -> Euler-angle attitude propagation
// Errors are a few ulp (validated against NumPy through the ginsim_rng_normals / ginsim_box_muller hooks:
// tests/test_gpu_parity.py::test_rng_words_bit_exact_and_normals, ::test_box_muller_corner_cases);
// the engine's parity tolerances are 1e-12..1e-9.
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
    double l[5];            // -2 log1p(-r'/2) = r' + r'^2 (1/4 + r'/12 + r'^2/32 + r'^3/80 + r'^4/192)
    double ln2_hi, ln2_lo;  // -2 ln 2, split
    double sc[5];           // sin: -1/3!, 1/5!, -1/7!, 1/9!, -1/11!   (Box-Muller uses 2, rotate_sincos 5)
    double cc[6];           // cos: -1/2!, 1/4!, ... 1/12!             (Box-Muller uses 2, rotate_sincos 6)
    double ang_bias, ang_scale;     // (0.5 - 2^14) 2 pi 2^-24 and 2 pi 2^-24: centred remainder of the 24-bit angle -> radians
    double u_hi, u_lo, u_half;      // 2^-32, 2^-40, 2^-41: uniform40 as two FMAs
    // OPAQUE = true pins the 23 constants in VGPRs (46 registers); false leaves them to the compiler (SGPR literals),
    // which is what the two-algorithm kernels need to stay under 256 VGPRs without scratch spills.
    template <bool OPAQUE>
    GINSIM_FM void init() {
        auto vconst = [](double x) { return OPAQUE ? ginsim::vconst(x) : x; };
        const double lc[5] = {0.25, 1.0 / 12.0, 1.0 / 32.0, 1.0 / 80.0, 1.0 / 192.0};
#pragma unroll
        for (int k = 0; k < 5; ++k) l[k] = vconst(lc[k]);
        ln2_hi = vconst(-2.0 * 6.93147180369123816490e-01);
        ln2_lo = vconst(-2.0 * 1.90821492927058770002e-10);
        const double s[5] = {-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0};
        const double c[6] = {-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0, 1.0 / 479001600.0};
#pragma unroll
        for (int k = 0; k < 5; ++k) sc[k] = vconst(s[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) cc[k] = vconst(c[k]);
        ang_bias = vconst((0.5 - 16384.0) * (6.283185307179586476925 * 0x1.0p-24));
        ang_scale = vconst(6.283185307179586476925 * 0x1.0p-24);
        u_hi = vconst(0x1.0p-32);
        u_lo = vconst(0x1.0p-40);
        u_half = vconst(0x1.0p-41);
    }
};

// Box-Muller lookup tables, built by every workgroup in LDS (12 KB):
//   lg[k] = {-2/c_k, -2 ln c_k}, c_k the (rounded) centre of the k-th of 256 mantissa bins of m in [sqrt(1/2), sqrt(2))
//           (c = 1 exactly for the bin that contains 1, so that ln u -> 0 without cancellation as u -> 1); the factor
//           -2 of the Box-Muller radius sqrt(-2 ln u) is folded into the table and the series;
//   sc[i] = {sin a_i, cos a_i}, a_i = 2 pi (i + 1/2) / 512: the centre of the i-th of 512 sectors of the turn.
// With them log needs a degree-6 series in |r| <= 2^-9 instead of a reciprocal, a quotient correction and a degree-21
// series, and sin/cos need two two-term series in |b| <= pi/512 and four FMAs instead of a quadrant reduction, two
// degree-15/16 series and the swap / sign selects.
constexpr int kLogBins = 256, kAngBins = 512;
struct NormalTables {
    const double2* lg;
    const double2* sc;
};

GINSIM_FM void fill_normal_tables(double2* tab, int tid, int nthreads) {
    for (int k = tid; k < kLogBins; k += nthreads) {
        const int h0 = (k << 12) + 0x3fe6a09e;
        const double m_lo = __hiloint2double(h0, 0), m_hi = __hiloint2double(h0 + 0x1000, 0);
        // ln c is taken for the ROUNDED reciprocal that is stored: m * (1/c) - 1 is then one exactly rounded fma of the
        // true ratio (a reciprocal rounded independently of ln c would put an ABSOLUTE 1e-16 on ln u and spoil the
        // relative accuracy of small |ln u|); the error of ln c scales with |ln c| ~ |ln u|.
        const bool unit = m_lo <= 1.0 && 1.0 < m_hi;
        const double inv = unit ? 1.0 : 1.0 / (0.5 * (m_lo + m_hi));
        // the unit bin adds 2^-200 instead of 0: absorbed by every other value, and u = 1 gives radius 2^-100, not 0/0
        tab[k] = double2{-2.0 * inv, unit ? 0x1.0p-200 : 2.0 * log(inv)};
    }
    for (int i = tid; i < kAngBins; i += nthreads) {
        double sn, cs;
        sincospi((double)(2 * i + 1) * (1.0 / kAngBins), &sn, &cs);
        tab[kLogBins + i] = double2{sn, cs};
    }
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

// sqrt(x) for finite x > 0 well inside the normal range (here x = -2 ln u in [2^-200, 75.5]: the log table returns
// 2^-200 instead of 0 for u = 1).  v_rsq_f64 estimate, one Goldschmidt step, one residual correction: <= 1 ulp,
// 6 VALU + v_rsq.  The compiler's sqrt() adds range scaling, a second correction and an inf/0 select (17 VALU).
GINSIM_FM double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);                                // ~2^-46
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);       // the 2^-23 error of h only scales the correction
}

// -2 ln u for 0 < u <= 1 (normal, not denormal: u >= 2^-41 by construction of uniform40): the squared Box-Muller radius.
//   u = m 2^e, m in [sqrt(1/2), sqrt(2));  -2 ln u = e (-2 ln2) + (-2 ln c_k) + (-2 log1p(r)),  r = m / c_k - 1,
//   |r| <= 2^-9;  with r' = -2 r = fma(m, -2/c_k, 2):  -2 log1p(r) = r' + r'^2/4 + r'^3/12 + r'^4/32 + r'^5/80 + r'^6/192
// The exponent/mantissa split and the bin index are integer arithmetic on the high word (no compare/select):
// adding (0x3ff00000 - 0x3fe6a09e) moves the sqrt(1/2) boundary onto an exponent boundary.
GINSIM_FM double neg2_log_u01(double u, const MathConsts& k, const NormalTables& tab) {
    uint32_t hx = (uint32_t)__double2hiint(u) + (0x3ff00000u - 0x3fe6a09eu);
    const int e = (int)(hx >> 20) - 0x3ff;
    const double2 t = tab.lg[(hx >> 12) & (kLogBins - 1)];
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    const double m = __hiloint2double((int)hx, __double2loint(u));
    const double r = __builtin_fma(m, t.x, 2.0);
    double p = k.l[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) p = __builtin_fma(p, r, k.l[i]);
    const double ed = (double)e;
    // (e (-2 ln2)_hi + (-2 ln c)) + (r' + e (-2 ln2)_lo + r'^2 p): the first sum is exact to 1 ulp
    const double small = __builtin_fma(r * r, p, __builtin_fma(ed, k.ln2_lo, r));
    return __builtin_fma(ed, k.ln2_hi, t.y) + small;
}

// sin and cos of the Box-Muller angle 2 pi (a + 1/2) 2^-24, a = the low 24 bits of a Philox word: sector i = top 9
// bits of a, b = centred remainder in radians (|b| <= pi/512), angle = a_i + b.  sin b = b + b^3 (-1/6 + b^2/120) (next
// term 6e-20), cos b - 1 = b^2 (-1/2 + b^2/24) (next term 7e-17).
GINSIM_FM void sincos_turn24(uint32_t w, double& s, double& c, const MathConsts& k, const NormalTables& tab) {
    const double2 t = tab.sc[(w >> 15) & (kAngBins - 1)];
    const double b = __builtin_fma((double)(w & 0x7fffu), k.ang_scale, k.ang_bias);        // centred remainder, radians
    const double tt = b * b;
    const double sb = __builtin_fma(b * tt, __builtin_fma(tt, k.sc[1], k.sc[0]), b);       // sin b
    const double cm = tt * __builtin_fma(tt, k.cc[1], k.cc[0]);                            // cos b - 1
    s = t.x + __builtin_fma(t.y, sb, t.x * cm);
    c = t.y + __builtin_fma(-t.x, sb, t.y * cm);
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
