// Lean fp64 elementary functions for the MC kernel, specialised to the argument ranges the kernel needs.
//
// ROCm's OCML double-precision log / sincos / sincospi are <1 ulp and pay for it with double-double
// arithmetic and a Payne-Hanek large-argument path (~1000 v_add_f64 in the first build of mc_kernel).
// The kernel only needs:
//   * log(u) for a uniform u in (0,1]               -> Box-Muller radius
//   * sin/cos(pi*x) for x = 2u in (0,2)             -> Box-Muller angle
//   * sin/cos(a + d) from sin/cos(a) for small |d|  -> Euler-angle attitude propagation
// Errors are a few ulp (validated against libm in tests/test_fastmath.py through the probe entry point);
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
    double lg[10];          // 1/3 .. 1/21 (atanh series)
    double ln2_hi, ln2_lo;
    double sc[7];           // sin: -1/3!, 1/5!, ... -1/15!
    double cc[8];           // cos: -1/2!, 1/4!, ... 1/16!
    double pio2_hi, pio2_lo;
    // OPAQUE = true pins the 29 constants in VGPRs (58 registers); false leaves them to the compiler (SGPR literals),
    // which is what the two-algorithm kernels need to stay under 256 VGPRs without scratch spills.
    template <bool OPAQUE>
    GINSIM_FM void init() {
        auto vconst = [](double x) { return OPAQUE ? ginsim::vconst(x) : x; };
#pragma unroll
        for (int k = 0; k < 10; ++k) lg[k] = vconst(1.0 / (2 * k + 3));
        ln2_hi = vconst(6.93147180369123816490e-01);
        ln2_lo = vconst(1.90821492927058770002e-10);
        const double s[7] = {-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0, 1.0 / 6227020800.0,
                             -1.0 / 1307674368000.0};
        const double c[8] = {-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0, 1.0 / 479001600.0,
                             -1.0 / 87178291200.0, 1.0 / 20922789888000.0};
#pragma unroll
        for (int k = 0; k < 7; ++k) sc[k] = vconst(s[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) cc[k] = vconst(c[k]);
        pio2_hi = vconst(1.57079632679489655800e+00);
        pio2_lo = vconst(6.12323399573676603587e-17);
    }
};

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

// sqrt(x) for finite x >= 0 well inside the normal range (here x = -2 ln u <= 75.5); x below 2^-200 (only u = 1) is
// lifted to 2^-200, i.e. returns 2^-100 instead of 0.  v_rsq_f64 estimate, one Goldschmidt step, one residual
// correction: <= 1 ulp.  The compiler's sqrt() adds range scaling, a second correction and an inf/0 select (17 VALU).
GINSIM_FM double sqrt_pos(double x) {
    x = __builtin_fmax(x, 0x1.0p-200);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}

// Natural log for 0 < u <= 1 (normal, not denormal: u >= 2^-54 by construction of uniform53).
//   u = m * 2^e, m in [sqrt(1/2), sqrt(2));  ln u = e ln2 + 2 atanh(s),  s = (m-1)/(m+1),  |s| <= 0.1716
// The exponent/mantissa split is done on the high word with integer arithmetic only (no compare/select):
// adding (0x3ff00000 - 0x3fe6a09e) moves the sqrt(1/2) boundary onto an exponent boundary.
GINSIM_FM double log_u01(double u) {
    uint32_t hx = (uint32_t)__double2hiint(u) + (0x3ff00000u - 0x3fe6a09eu);
    const int e = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    const double m = __hiloint2double((int)hx, __double2loint(u));
    const double num = m - 1.0, den = m + 1.0;
    const double y = rcp_n1(den);
    double s = num * y;
    s = __builtin_fma(__builtin_fma(-den, s, num), y, s);   // one correction step: s = num/den to ~1 ulp
    const double t = s * s;
    // atanh(s)/s = sum t^k/(2k+1); t <= 0.02944 -> k = 10 leaves 2e-17
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, t, 1.0 / 19.0);
    p = __builtin_fma(p, t, 1.0 / 17.0);
    p = __builtin_fma(p, t, 1.0 / 15.0);
    p = __builtin_fma(p, t, 1.0 / 13.0);
    p = __builtin_fma(p, t, 1.0 / 11.0);
    p = __builtin_fma(p, t, 1.0 / 9.0);
    p = __builtin_fma(p, t, 1.0 / 7.0);
    p = __builtin_fma(p, t, 1.0 / 5.0);
    p = __builtin_fma(p, t, 1.0 / 3.0);
    const double s2 = s + s;
    const double lnm = __builtin_fma(s2 * t, p, s2);          // 2s + 2s t p
    const double ed = (double)e;
    // ln2 split so that ed * ln2_hi is exact for |e| < 2^10
    return __builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, lnm));
}

// sin/cos of theta for |theta| <= pi/4 (Taylor, truncation < 5e-17 / 2e-18)
GINSIM_FM void sincos_q(double th, double& s, double& c) {
    const double t = th * th;
    double ps = -1.0 / 1307674368000.0;                        // -1/15!
    ps = __builtin_fma(ps, t, 1.0 / 6227020800.0);             //  1/13!
    ps = __builtin_fma(ps, t, -1.0 / 39916800.0);              // -1/11!
    ps = __builtin_fma(ps, t, 1.0 / 362880.0);                 //  1/9!
    ps = __builtin_fma(ps, t, -1.0 / 5040.0);
    ps = __builtin_fma(ps, t, 1.0 / 120.0);
    ps = __builtin_fma(ps, t, -1.0 / 6.0);
    s = __builtin_fma(th * t, ps, th);
    double pc = 1.0 / 20922789888000.0;                        //  1/16!
    pc = __builtin_fma(pc, t, -1.0 / 87178291200.0);           // -1/14!
    pc = __builtin_fma(pc, t, 1.0 / 479001600.0);              //  1/12!
    pc = __builtin_fma(pc, t, -1.0 / 3628800.0);               // -1/10!
    pc = __builtin_fma(pc, t, 1.0 / 40320.0);
    pc = __builtin_fma(pc, t, -1.0 / 720.0);
    pc = __builtin_fma(pc, t, 1.0 / 24.0);
    pc = __builtin_fma(pc, t, -0.5);
    c = __builtin_fma(t, pc, 1.0);
}

// sin(pi x), cos(pi x) for 0 <= x <= 2.  2x = k + r, k integer, |r| <= 1/2 (exact), angle = k pi/2 + r pi/2.
// The argument is x2 = 2x, the angle in quarter turns (what uniform53q delivers).
GINSIM_FM void sincos_quarters(double x2, double& s, double& c) {
    const double kd = __builtin_rint(x2);
    const double r = x2 - kd;                                   // exact
    const int k = (int)kd;
    // theta = r * pi/2 with a two-term constant so the product carries ~1 ulp of the angle
    const double th = __builtin_fma(r, 1.57079632679489655800e+00, r * 6.12323399573676603587e-17);
    double sq, cq;
    sincos_q(th, sq, cq);
    const bool swap = (k & 1) != 0;
    const double ss = swap ? cq : sq;
    const double cc = swap ? sq : cq;
    // negate through the sign bit: sin flips for k in {2,3}, cos for k in {1,2} (mod 4)
    s = __hiloint2double(__double2hiint(ss) ^ ((k & 2) << 30), __double2loint(ss));
    c = __hiloint2double(__double2hiint(cc) ^ (((k + 1) & 2) << 30), __double2loint(cc));
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

// ---- the same functions with their coefficients taken from VGPR-resident constants (hot loop) -------------
GINSIM_FM double log_u01(double u, const MathConsts& k) {
    uint32_t hx = (uint32_t)__double2hiint(u) + (0x3ff00000u - 0x3fe6a09eu);
    const int e = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    const double m = __hiloint2double((int)hx, __double2loint(u));
    const double num = m - 1.0, den = m + 1.0;
    const double y = rcp_n1(den);
    double s = num * y;
    s = __builtin_fma(__builtin_fma(-den, s, num), y, s);
    const double t = s * s;
    double p = k.lg[9];
#pragma unroll
    for (int i = 8; i >= 0; --i) p = __builtin_fma(p, t, k.lg[i]);
    const double s2 = s + s;
    const double lnm = __builtin_fma(s2 * t, p, s2);
    const double ed = (double)e;
    return __builtin_fma(ed, k.ln2_hi, __builtin_fma(ed, k.ln2_lo, lnm));
}

GINSIM_FM void sincos_quarters(double x2, double& s, double& c, const MathConsts& k) {
    const double kd = __builtin_rint(x2);
    const double r = x2 - kd;
    const int q = (int)kd;
    const double th = __builtin_fma(r, k.pio2_hi, r * k.pio2_lo);
    const double t = th * th;
    double ps = k.sc[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) ps = __builtin_fma(ps, t, k.sc[i]);
    const double sq = __builtin_fma(th * t, ps, th);
    double pc = k.cc[7];
#pragma unroll
    for (int i = 6; i >= 0; --i) pc = __builtin_fma(pc, t, k.cc[i]);
    const double cq = __builtin_fma(t, pc, 1.0);
    const bool swap = (q & 1) != 0;
    const double ss = swap ? cq : sq;
    const double cs = swap ? sq : cq;
    s = __hiloint2double(__double2hiint(ss) ^ ((q & 2) << 30), __double2loint(ss));
    c = __hiloint2double(__double2hiint(cs) ^ (((q + 1) & 2) << 30), __double2loint(cs));
}

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

}  // namespace ginsim
