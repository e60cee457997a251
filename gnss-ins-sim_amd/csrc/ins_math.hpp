// Strapdown building blocks of the MC kernels.
//
// Restates the ZYX subset of gnss_ins_sim/attitude/attitude.py and the WGS-84 model of
// gnss_ins_sim/geoparams/geoparams.py as register-resident state: the attitude carries its own
// sin/cos so that the three sincos evaluated for euler2dcm(att[i]) (attitude.py:344-371) are reused
// by euler_update_zyx at step i+1 (attitude.py:679-721) -- 3 sincos per step instead of 7 trig calls.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "fastmath.hpp"

namespace ginsim {

#define GINSIM_HD __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kHalfPi = 0.5 * kPi;
// geoparams.py:17-23, 40-43
constexpr double kRe = 6378137.0;
constexpr double kFlat = 1.0 / 298.257223563;
constexpr double kEcc = 0.0818191908426215;
constexpr double kEsq = kEcc * kEcc;
constexpr double kWie = 7292115e-11;
constexpr double kG0 = 9.7803253359;
constexpr double kGk = 0.00193185265241;
constexpr double kGm = 0.00344978650684;
constexpr int kTrigResync = 32;   // exact sincos of the attitude (and latitude) every this many steps

struct Vec3 { double x, y, z; };

GINSIM_HD Vec3 cross3(const Vec3& a, const Vec3& b) {   // attitude.cross3, attitude.py:758-770
    return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Geo { double rm, rn, g, sl, cl; };

// geoparams.geo_param, geoparams.py:25-53, with sin/cos(lat) supplied by the caller.  The three divisions by
// sqrt(1 - e^2 sin^2) become one reciprocal square root (Newton-refined v_rsq_f64) and products.
GINSIM_HD Geo geo_param_sc(double sl, double cl, double h) {
    Geo o;
    o.sl = sl;
    o.cl = cl;
    const double s2 = sl * sl;
    const double q = 1.0 - kEsq * s2;
    const double iw = rsqrt_nr(q);
    o.rn = kRe * iw;
    o.rm = (kRe * (1.0 - kEsq)) * iw * (iw * iw);
    const double g1 = kG0 * (1.0 + kGk * s2) * iw;
    o.g = g1 * (1.0 - (2.0 / kRe) * (1.0 + kFlat + kGm - 2.0 * kFlat * s2) * h + (3.0 / (kRe * kRe)) * (h * h));
    return o;
}

GINSIM_HD Geo geo_param(double lat, double h) {
    double sl, cl;
    sincos(lat, &sl, &cl);
    return geo_param_sc(sl, cl, h);
}

// geoparams.lla2ecef, geoparams.py:70-87
GINSIM_HD Vec3 lla2ecef(double lat, double lon, double alt) {
    const double sl = sin(lat), cl = cos(lat);
    const double r = kRe / sqrt(1.0 - kEsq * sl * sl);
    const double rho = (r + alt) * cl;
    return Vec3{rho * cos(lon), rho * sin(lon), (r * (1.0 - kEsq) + alt) * sl};
}

// LLA error -> local NED metres: array_error(lla=1), ins_data_manager.py:542-552 with attitude.ecef_to_ned
// (attitude.py:596-603): err = C_ne(ref lat, lon) . (lla2ecef(x) - lla2ecef(ref)).
GINSIM_HD Vec3 lla_error_ned(const Vec3& x, const Vec3& ref) {
    const Vec3 a = lla2ecef(x.x, x.y, x.z), b = lla2ecef(ref.x, ref.y, ref.z);
    const Vec3 d{a.x - b.x, a.y - b.y, a.z - b.z};
    double sl, cl, so, co;
    sincos(ref.x, &sl, &cl);
    sincos(ref.y, &so, &co);
    return Vec3{-sl * co * d.x - sl * so * d.y + cl * d.z,
                -so * d.x + co * d.y,
                -cl * co * d.x - cl * so * d.y - sl * d.z};
}

// ZYX Euler attitude with cached trig.
struct Att {
    double yaw, pit, rol;
    double sy, cy, sp, cp, sr, cr;

    GINSIM_HD void set(double y, double p, double r) {
        yaw = y; pit = p; rol = r;
        sincos(y, &sy, &cy);
        sincos(p, &sp, &cp);
        sincos(r, &sr, &cr);
    }
    // attitude.euler2dcm(.,'zyx') rows, attitude.py:360-368 (n -> b)
    GINSIM_HD Vec3 to_body(const Vec3& v) const {
        return Vec3{cp * cy * v.x + cp * sy * v.y - sp * v.z,
                    (sr * sp * cy - cr * sy) * v.x + (sr * sp * sy + cr * cy) * v.y + cp * sr * v.z,
                    (sp * cr * cy + sy * sr) * v.x + (sp * cr * sy - cy * sr) * v.y + cp * cr * v.z};
    }
    GINSIM_HD Vec3 to_nav(const Vec3& v) const {   // transpose
        return Vec3{cp * cy * v.x + (sr * sp * cy - cr * sy) * v.y + (sp * cr * cy + sy * sr) * v.z,
                    cp * sy * v.x + (sr * sp * sy + cr * cy) * v.y + (sp * cr * sy - cy * sr) * v.z,
                    -sp * v.x + cp * sr * v.y + cp * cr * v.z};
    }
    // third column of the n->b DCM: C . [0,0,1]
    GINSIM_HD Vec3 down_in_body() const { return Vec3{-sp, cp * sr, cp * cr}; }
    // first row of the n->b DCM: C^T . [1,0,0]
    GINSIM_HD Vec3 fwd_in_nav() const { return Vec3{cp * cy, cp * sy, -sp}; }

    // attitude.euler_update_zyx, attitude.py:679-721, then refresh the cached trig.
    // The angles themselves are propagated exactly as the reference does; their cached sin/cos are ROTATED by
    // the step (angle-addition with short sin/cos(d) series, fastmath.hpp) instead of re-evaluated, and are
    // re-evaluated exactly when `resync` is set (every kTrigResync steps, wave-uniform), when the pitch folds
    // over +-pi/2, or when a step exceeds 0.25 rad.  +-2pi wraps leave the trig untouched.  A lane that steps by no
    // more than 2^-6 rad uses the three-term series.
    // EASY = true (the one-wavefront-per-run-group kernels in generate mode): ONE wave-uniform test in front of everything that is rare.  A lane
    // is "easy" when it needs neither the exact trig, nor the pitch fold, nor a +-2 pi wrap, and steps by no more than 2^-6 rad
    // -- every sample of a vehicle profile at >= 100 Hz; when all lanes of the wavefront are easy the step is three short
    // rotations and no exec-mask juggling (the per-lane branches below are ~45 scalar instructions per step).  The general
    // path does, lane by lane, exactly the same operations on an easy lane, so a run's bits do not depend on its wavefront
    // neighbours.  Measured (round 3): C3-shaped launches -4.8 % (end-point only) / -2.8 % (online statistics); the given-sensors
    // kernel, which waits for memory, does not gain when it materialises (+2 %) and stays as it was; the
    // wave-specialised kernels do not gain (their consumer sits at the 168-register limit: 12-52 B of scratch with it).
    template <bool EASY = false>
    GINSIM_HD void step(const Vec3& w, double dt, bool resync, const MathConsts& mk) {
        const double q = w.z * cr + w.y * sr;
        const double icp = rcp_n1(cp);      // 2^-46 relative on a rate that is multiplied by dt: far below the state's ulp
        const double dy = q * icp * dt;
        const double dp = (w.y * cr - w.z * sr) * dt;
        const double dr = (w.x + q * (sp * icp)) * dt;
        double y = yaw + dy;
        double p = pit + dp;
        double r = rol + dr;
        const bool fold = (p > kHalfPi) || (p < -kHalfPi);
        const double big = fmax(fabs(dy), fmax(fabs(dp), fabs(dr)));
        if (EASY) {
            const bool easy = !resync && !fold && (big <= 0x1.0p-6) && !(y > kPi) && !(y < -kPi) && !(r > kPi) && !(r < -kPi);
            if (__builtin_amdgcn_ballot_w64(!easy) == 0) {
                rotate_sincos_small(dy, sy, cy, mk);
                rotate_sincos_small(dp, sp, cp, mk);
                rotate_sincos_small(dr, sr, cr, mk);
                yaw = y; pit = p; rol = r;
                return;
            }
        }
        if (resync || fold || !(big <= 0.25)) {
            if (p > kHalfPi) {
                p = kPi - p; y += kPi; r += kPi;
            } else if (p < -kHalfPi) {
                p = -kPi - p; y += kPi; r += kPi;
            }
            if (y > kPi) y -= kTwoPi; else if (y < -kPi) y += kTwoPi;
            if (r > kPi) r -= kTwoPi; else if (r < -kPi) r += kTwoPi;
            set(y, p, r);
        } else {
            // Which series rotates the cached trig is decided PER LANE from the lane's own step, so a run's bits never
            // depend on which other runs share its wavefront (any sharding of the runs over launches / GPUs reproduces
            // them exactly).  When all lanes agree -- every vehicle profile at >= 100 Hz -- the other side of the
            // branch is skipped (s_cbranch_execz): measured no slower than a wave-wide vote.
            if (big <= 0x1.0p-6) {
                rotate_sincos_small(dy, sy, cy, mk);
                rotate_sincos_small(dp, sp, cp, mk);
                rotate_sincos_small(dr, sr, cr, mk);
            } else {
                rotate_sincos(dy, sy, cy, mk);
                rotate_sincos(dp, sp, cp, mk);
                rotate_sincos(dr, sr, cr, mk);
            }
            if (y > kPi) y -= kTwoPi; else if (y < -kPi) y += kTwoPi;
            if (r > kPi) r -= kTwoPi; else if (r < -kPi) r += kTwoPi;
            yaw = y; pit = p; rol = r;
        }
    }
};

// attitude.angle_range_pi, attitude.py:799-812 (Python float %: result carries the divisor's sign)
GINSIM_HD double angle_range_pi(double x) {
    double m = x - kTwoPi * floor(x / kTwoPi);
    if (m >= kTwoPi) m -= kTwoPi;
    if (m < 0.0) m += kTwoPi;
    return m > kPi ? m - kTwoPi : m;
}

}  // namespace ginsim
