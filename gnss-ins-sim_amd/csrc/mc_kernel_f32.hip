// fp32 variant of the fused Monte-Carlo kernel (BASELINE config 5).  Same algorithm, same launch geometry and the same
// SoA [component][sample][run] outputs as mc_kernel.hip, in single precision -- and, since round 3, DEFINED operation by
// operation, so that the float restatement in the test oracle (oracle_mc_run_f32 under oracle/c/) reproduces the sensor series
// and the trajectories of every run TO THE BIT:
//
//   * noise: exactly the normals of the fp64 path (philox.hpp: Philox4x32-7, three blocks per IMU step, the normal
//     transform that is defined in IEEE single precision) -- the fp64 kernel widens those floats, this one uses them as
//     they are.  Identical seeds therefore give the SAME noise realisation in both precisions, and the fp32 / fp64
//     trajectories of a run differ by rounding only (tests/test_gpu_fp32.py states the tolerances);
//   * every product, sum, quotient and square root below is ONE IEEE single-precision operation in source order: the file
//     is compiled with -ffp-contract=off, fused multiply-adds appear only where __builtin_fmaf spells them, quotients are
//     the correctly rounded `/` (no v_rcp_f32 / v_rsq_f32 approximations), and the exact sin/cos of the attitude
//     (initial state, every kTrigResync steps, pitch fold, steps > 0.25 rad) come from sincos_def(): quadrant reduction
//     and Taylor polynomials evaluated in fp64 with explicit fused multiply-adds, rounded to float once;
//   * attitude, body/NED velocity and all sensor arithmetic in fp32; the running sums (three Euler angles, three
//     velocity components) are Kahan-compensated so that 1e5 forward-Euler steps do not random-walk in the last bit;
//   * POSITION is accumulated in fp64 (3 adds per step): ECEF (4.7e6 m) and LLA radians do not fit fp32 (ulp 0.5 m /
//     6e-8 rad).  ref_frame 1 accumulates the DISPLACEMENT from the initial position (the series written to HBM is that
//     sum rounded to float, independent of how lla2ecef rounds the origin); ref_frame 0 accumulates lat / lon / alt and
//     writes pos - pos0 rounded to float.
//
// Restates the same reference functions as mc_kernel.hip (pathgen.py:441-594, free_integration.py:63-174,
// free_integration_odo.py:63-160, ins_data_manager.py:537-541).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "ginsim.h"
#include "ins_math.hpp"
#include "philox.hpp"
#include "device_once.hpp"

namespace ginsim {

namespace f32 {

#pragma clang fp contract(off)

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kTwoPiHi = 6.28318548202514648438f;        // float(2 pi)
constexpr float kTwoPiLo = -1.74845553146951715e-07f;      // 2 pi - float(2 pi)
constexpr float kHalfPiF = 1.57079632679489661923f;

#define F32_FM __device__ __forceinline__
F32_FM float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct V3 { float x, y, z; };

// a x b, each component one product and one fused multiply-add: a.y b.z - a.z b.y = fma(a.y, b.z, -(a.z b.y))
F32_FM V3 cross(const V3& a, const V3& b) {
    return V3{fm(a.y, b.z, -(a.z * b.y)), fm(a.z, b.x, -(a.x * b.z)), fm(a.x, b.y, -(a.y * b.x))};
}

// compensated accumulator: value += inc with the rounding error carried in comp
struct Acc {
    float v, comp;
    F32_FM void add(float inc) {
        const float t = inc - comp;
        const float s = v + t;
        comp = (s - v) - t;
        v = s;
    }
};

// sin and cos of x (|x| up to a few turns: Euler angles, latitude), defined in fp64 and rounded to float once:
//   k = rint(x 2/pi);  r = x - k pi/2 (two fused steps, pi/2 split in two doubles);  t = r^2
//   sin r = r + r t (S1 + t (S2 + t (S3 + t (S4 + t S5))))          |r| <= pi/4: truncation 7e-12
//   cos r = 1 + t (C1 + t (C2 + t (C3 + t (C4 + t (C5 + t C6)))))   truncation 1e-13
// then the quadrant k mod 4 swaps / negates.  The test oracle (sincos_def under oracle/c/) repeats these operations.
F32_FM void sincos_def(double x, float& sn, float& cs) {
    const double k = __builtin_rint(x * 0.63661977236758134308);
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    const double t = r * r;
    double ps = -1.0 / 39916800.0;
    ps = __builtin_fma(ps, t, 1.0 / 362880.0);
    ps = __builtin_fma(ps, t, -1.0 / 5040.0);
    ps = __builtin_fma(ps, t, 1.0 / 120.0);
    ps = __builtin_fma(ps, t, -1.0 / 6.0);
    const double s = __builtin_fma(r * t, ps, r);
    double pc = 1.0 / 479001600.0;
    pc = __builtin_fma(pc, t, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, t, 1.0 / 40320.0);
    pc = __builtin_fma(pc, t, -1.0 / 720.0);
    pc = __builtin_fma(pc, t, 1.0 / 24.0);
    pc = __builtin_fma(pc, t, -0.5);
    const double c = __builtin_fma(t, pc, 1.0);
    const int q = (int)k & 3;
    const double so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    sn = (float)((q & 2) ? -so : so);
    cs = (float)(((q + 1) & 2) ? -co : co);
}

F32_FM void rotate(float d, float& s, float& c) {      // sin/cos(a+d) from sin/cos(a), |d| <= 0.25
    const float t = d * d;
    const float sd = d * fm(t, fm(t, 8.3333333e-3f, -1.6666667e-1f), 1.0f);
    const float cm1 = t * fm(t, fm(t, -1.3888889e-3f, 4.1666667e-2f), -0.5f);
    const float s0 = s, c0 = c;
    s = fm(c0, sd, fm(s0, cm1, s0));
    c = fm(-s0, sd, fm(c0, cm1, c0));
}

struct Att {
    Acc yaw, pit, rol;
    float sy, cy, sp, cp, sr, cr;
    F32_FM void resync() {
        sincos_def((double)yaw.v, sy, cy);
        sincos_def((double)pit.v, sp, cp);
        sincos_def((double)rol.v, sr, cr);
    }
    F32_FM void set(float y, float p, float r) {
        yaw = Acc{y, 0.f}; pit = Acc{p, 0.f}; rol = Acc{r, 0.f};
        resync();
    }
    // rows of attitude.euler2dcm(.,'zyx') (attitude.py:360-368, n -> b): every entry one product, or one product and one
    // fused multiply-add; a matrix-vector product is one product and two fused multiply-adds per component
    F32_FM V3 to_body(const V3& v) const {
        const float srsp = sr * sp, spcr = sp * cr;
        const float c11 = cp * cy, c12 = cp * sy, c13 = -sp;
        const float c21 = fm(srsp, cy, -(cr * sy)), c22 = fm(srsp, sy, cr * cy), c23 = cp * sr;
        const float c31 = fm(spcr, cy, sy * sr), c32 = fm(spcr, sy, -(cy * sr)), c33 = cp * cr;
        return V3{fm(c13, v.z, fm(c12, v.y, c11 * v.x)), fm(c23, v.z, fm(c22, v.y, c21 * v.x)), fm(c33, v.z, fm(c32, v.y, c31 * v.x))};
    }
    F32_FM V3 to_nav(const V3& v) const {
        const float srsp = sr * sp, spcr = sp * cr;
        const float c11 = cp * cy, c12 = cp * sy, c13 = -sp;
        const float c21 = fm(srsp, cy, -(cr * sy)), c22 = fm(srsp, sy, cr * cy), c23 = cp * sr;
        const float c31 = fm(spcr, cy, sy * sr), c32 = fm(spcr, sy, -(cy * sr)), c33 = cp * cr;
        return V3{fm(c31, v.z, fm(c21, v.y, c11 * v.x)), fm(c32, v.z, fm(c22, v.y, c12 * v.x)), fm(c33, v.z, fm(c23, v.y, c13 * v.x))};
    }
    F32_FM V3 down_in_body() const { return V3{-sp, cp * sr, cp * cr}; }
    F32_FM V3 fwd_in_nav() const { return V3{cp * cy, cp * sy, -sp}; }

    // attitude.euler_update_zyx (attitude.py:679-721) in fp32
    F32_FM void step(const V3& w, float dt, bool do_resync) {
        const float q = fm(w.z, cr, w.y * sr);
        const float icp = 1.0f / cp;
        const float dy = (q * icp) * dt;
        const float dp = fm(w.y, cr, -(w.z * sr)) * dt;
        const float dr = fm(q, sp * icp, w.x) * dt;
        yaw.add(dy); pit.add(dp); rol.add(dr);
        const float big = fmaxf(fabsf(dy), fmaxf(fabsf(dp), fabsf(dr)));
        // The common step is three rotations of the cached sines / cosines and nothing else.  ONE wave-uniform branch, taken
        // AFTER them, for everything that is rare (exact trig every kTrigResync steps, pitch over the pole, yaw / roll over
        // +-pi, a step > 0.25 rad): it overwrites what the rotations left, so the common path carries no copies to a merge
        // point.  A lane that did not need the branch keeps its rotated values inside it: a run's bits do not depend on
        // its wavefront neighbours.
        rotate(dy, sy, cy); rotate(dp, sp, cp); rotate(dr, sr, cr);
        const bool rare = do_resync || !(fabsf(pit.v) <= kHalfPiF) || !(fabsf(yaw.v) <= kPiF) || !(fabsf(rol.v) <= kPiF) || !(big <= 0.25f);
        if (__builtin_amdgcn_ballot_w64(rare) != 0) {
            const bool fold = (pit.v > kHalfPiF) || (pit.v < -kHalfPiF);
            if (fold) {
                pit = Acc{pit.v > 0.f ? kPiF - pit.v : -kPiF - pit.v, 0.f};
                yaw.add(kPiF); rol.add(kPiF);
            }
            if (yaw.v > kPiF) { yaw.add(-kTwoPiHi); yaw.add(-kTwoPiLo); } else if (yaw.v < -kPiF) { yaw.add(kTwoPiHi); yaw.add(kTwoPiLo); }
            if (rol.v > kPiF) { rol.add(-kTwoPiHi); rol.add(-kTwoPiLo); } else if (rol.v < -kPiF) { rol.add(kTwoPiHi); rol.add(kTwoPiLo); }
            if (do_resync || fold || !(big <= 0.25f)) resync();
        }
    }
};

struct Nav {
    Att att;
    Acc vb[3];      // body velocity (ref_frame 1)
    Acc vn[3];      // NED velocity state (ref_frame 0)
    V3 vel;         // navigation-frame velocity of the previous sample
    double pos[3];  // fp64: displacement from the initial position (ref_frame 1) / lat, lon, alt (ref_frame 0)
    double pos0[3]; // the initial position: ECEF (ref_frame 1, enters the end-point error only) / lat, lon, alt
    float g;
    float sl, cl;   // sin/cos latitude (ref_frame 0), rotated per step, exact every resync
    bool ext_g;
};

template <int RF>
F32_FM void nav_init(Nav& s, const double* __restrict__ ini, int has_g) {
    s.att.set((float)ini[6], (float)ini[7], (float)ini[8]);
    const V3 vb{(float)ini[3], (float)ini[4], (float)ini[5]};
    s.vb[0] = Acc{vb.x, 0.f}; s.vb[1] = Acc{vb.y, 0.f}; s.vb[2] = Acc{vb.z, 0.f};
    s.vel = s.att.to_nav(vb);
    s.vn[0] = Acc{s.vel.x, 0.f}; s.vn[1] = Acc{s.vel.y, 0.f}; s.vn[2] = Acc{s.vel.z, 0.f};
    if (RF == 1) {
        const Vec3 e = lla2ecef(ini[0], ini[1], ini[2]);
        s.pos0[0] = e.x; s.pos0[1] = e.y; s.pos0[2] = e.z;
        s.pos[0] = 0.0; s.pos[1] = 0.0; s.pos[2] = 0.0;
        s.g = (float)(has_g ? ini[9] : geo_param(ini[0], ini[2]).g);
    } else {
        s.pos0[0] = s.pos[0] = ini[0]; s.pos0[1] = s.pos[1] = ini[1]; s.pos0[2] = s.pos[2] = ini[2];
        s.g = has_g ? (float)ini[9] : 0.f;
    }
    sincos_def(ini[0], s.sl, s.cl);
    s.ext_g = has_g != 0;
}

template <int RF, bool ODO>
F32_FM void nav_step(Nav& s, const V3& gyro, const V3& accel, float odo, float dt, int earth_rot, bool resync) {
    if (RF == 1) {
        const V3 v_prev = s.vel;
        if (!ODO) {
            const V3 gb = s.att.down_in_body();
            const V3 vb{s.vb[0].v, s.vb[1].v, s.vb[2].v};
            const V3 wxv = cross(gyro, vb);
            s.vb[0].add((fm(gb.x, s.g, accel.x) - wxv.x) * dt);
            s.vb[1].add((fm(gb.y, s.g, accel.y) - wxv.y) * dt);
            s.vb[2].add((fm(gb.z, s.g, accel.z) - wxv.z) * dt);
        }
        s.att.step(gyro, dt, resync);
        if (ODO) {
            const V3 f = s.att.fwd_in_nav();
            s.vel = V3{f.x * odo, f.y * odo, f.z * odo};
        } else {
            s.vel = s.att.to_nav(V3{s.vb[0].v, s.vb[1].v, s.vb[2].v});
        }
        s.pos[0] += (double)(v_prev.x * dt);
        s.pos[1] += (double)(v_prev.y * dt);
        s.pos[2] += (double)(v_prev.z * dt);
    } else {
        // geoparams.geo_param (geoparams.py:25-53) in fp32 on the cached sin/cos(lat); altitude from the fp64 state
        const float h = (float)s.pos[2];
        const float s2 = s.sl * s.sl;
        const float qq = fm(-(float)kEsq, s2, 1.0f);
        const float sq = __builtin_sqrtf(qq);
        const float rn = (float)kRe / sq;
        const float rm = (float)(kRe * (1.0 - kEsq)) / (qq * sq);
        const float g1 = ((float)kG0 * fm((float)kGk, s2, 1.0f)) / sq;
        const float gh = fm((float)(3.0 / (kRe * kRe)), h * h, fm(-((float)(2.0 / kRe) * fm(-2.0f * (float)kFlat, s2, (float)(1.0 + kFlat + kGm))), h, 1.0f));
        const float gm = g1 * gh;
        const float irm = 1.0f / (rm + h), irn = 1.0f / (rn + h), icl = 1.0f / s.cl;
        const V3 v = s.vel;
        const V3 w_en{v.y * irn, -(v.x * irm), -(((v.y * s.sl) * icl) * irn)};
        V3 w_ie{0.f, 0.f, 0.f};
        if (earth_rot) { w_ie.x = (float)kWie * s.cl; w_ie.z = -((float)kWie * s.sl); }
        const V3 wb = s.att.to_body(V3{w_en.x + w_ie.x, w_en.y + w_ie.y, w_en.z + w_ie.z});
        const V3 w_nb{gyro.x - wb.x, gyro.y - wb.y, gyro.z - wb.z};
        if (!ODO) {
            const V3 an = s.att.to_nav(accel);
            const float g = s.ext_g ? s.g : gm;
            const V3 cor = cross(V3{fm(2.f, w_ie.x, w_en.x), fm(2.f, w_ie.y, w_en.y), fm(2.f, w_ie.z, w_en.z)}, v);
            s.vn[0].add((an.x - cor.x) * dt);
            s.vn[1].add((an.y - cor.y) * dt);
            s.vn[2].add(((an.z + g) - cor.z) * dt);
        }
        s.att.step(w_nb, dt, resync);
        const float dlat = (v.x * irm) * dt;
        s.pos[0] += (double)dlat;
        s.pos[1] += (double)(((v.y * irn) * icl) * dt);
        s.pos[2] += (double)(-(v.z * dt));
        if (resync) sincos_def(s.pos[0], s.sl, s.cl);
        else rotate(dlat, s.sl, s.cl);
        if (ODO) {
            const V3 f = s.att.fwd_in_nav();
            s.vel = V3{f.x * odo, f.y * odo, f.z * odo};
        } else {
            s.vel = V3{s.vn[0].v, s.vn[1].v, s.vn[2].v};
        }
    }
}

// written once, never read back by the kernel: non-temporal stores
F32_FM void st(float* p, float v) { __builtin_nontemporal_store(v, p); }

template <int RF>
F32_FM void store9(float* __restrict__ base, int64_t plane, int64_t off, const Nav& s) {
    st(base + 0 * plane + off, s.att.yaw.v);
    st(base + 1 * plane + off, s.att.pit.v);
    st(base + 2 * plane + off, s.att.rol.v);
    if (RF == 1) {      // the state IS the displacement from the initial position
        st(base + 3 * plane + off, (float)s.pos[0]);
        st(base + 4 * plane + off, (float)s.pos[1]);
        st(base + 5 * plane + off, (float)s.pos[2]);
    } else {
        st(base + 3 * plane + off, (float)(s.pos[0] - s.pos0[0]));
        st(base + 4 * plane + off, (float)(s.pos[1] - s.pos0[1]));
        st(base + 5 * plane + off, (float)(s.pos[2] - s.pos0[2]));
    }
    st(base + 6 * plane + off, s.vel.x);
    st(base + 7 * plane + off, s.vel.y);
    st(base + 8 * plane + off, s.vel.z);
}

typedef const ginsim_mc_params __attribute__((address_space(4))) * params_ptr;
typedef const double __attribute__((address_space(4))) * uniform_ptr;

template <int RF>
F32_FM void store_end(double* __restrict__ out, int64_t runs, int64_t r, const Nav& s) {
    params_ptr kp = (params_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    out[0 * runs + r] = angle_range_pi((double)s.att.yaw.v - kp->ref_end[0]);
    out[1 * runs + r] = angle_range_pi((double)s.att.pit.v - kp->ref_end[1]);
    out[2 * runs + r] = angle_range_pi((double)s.att.rol.v - kp->ref_end[2]);
    const Vec3 p = RF == 1 ? Vec3{s.pos0[0] + s.pos[0], s.pos0[1] + s.pos[1], s.pos0[2] + s.pos[2]} : Vec3{s.pos[0], s.pos[1], s.pos[2]};
    Vec3 ep{p.x - kp->ref_end[3], p.y - kp->ref_end[4], p.z - kp->ref_end[5]};
    if (kp->end_pos_ned && kp->ref_frame == 0) ep = lla_error_ned(p, Vec3{kp->ref_end[3], kp->ref_end[4], kp->ref_end[5]});
    out[3 * runs + r] = ep.x;
    out[4 * runs + r] = ep.y;
    out[5 * runs + r] = ep.z;
    out[6 * runs + r] = (double)s.vel.x - kp->ref_end[6];
    out[7 * runs + r] = (double)s.vel.y - kp->ref_end[7];
    out[8 * runs + r] = (double)s.vel.z - kp->ref_end[8];
}

struct Model { float bias[3], a[3], b[3], w[3]; int wd[3]; };

F32_FM void load_model(const ginsim_sensor_model& m, Model& o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.bias[i] = (float)m.bias[i]; o.a[i] = (float)m.gm_a[i]; o.b[i] = (float)m.gm_b[i]; o.w[i] = (float)m.white[i];
        o.wd[i] = m.white_drift[i];
    }
}

// sensor sample: float(truth) [+ bias] + drift, then the white noise as one fused multiply-add; Gauss-Markov update as
// one product and one fused multiply-add (pathgen.py:500, 562, 589-590).  WD = false: no white-drift axis and no constant
// bias (every standard IMU grade) -- the selects and the bias additions are compiled out (x + 0 == x).
template <bool WD>
F32_FM V3 sense3(const double (&truth)[3], const Model& m, float (&drift)[3], const float* zd, const float* zw) {
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float bz = m.b[i] * zd[i];
        const float d = (WD && m.wd[i]) ? bz : drift[i];
        const float base = WD ? ((float)truth[i] + m.bias[i]) + d : (float)truth[i] + d;
        o[i] = fm(m.w[i], zw[i], base);
        drift[i] = fm(m.a[i], drift[i], bz);
    }
    return V3{o[0], o[1], o[2]};
}

// The vibration term of Sim(env=...) in single precision (ABI 5; pathgen.py:476-492, 538-556), added last as the reference's sum
// does and DEFINED like the rest of this file: the amplitudes rounded to float once; 'random': o = fma(amp, z, o) with the
// normals of streams STREAM, STREAM + 1 (one Philox block); 'sinusoidal': the angle omega_dt * j (+ phase) in fp64 -- one
// product, one sum --, its sine by sincos_def (fp64 polynomial, rounded to float once), o = fma(amp, sin, o).
struct Vib { int type, random_phase; float amp[3]; double omega_dt, phase[3]; };

template <uint32_t PHASE_STREAM>
F32_FM void load_vib(const ginsim_vibration& v, const RngKey& key, Vib& o) {
    o.type = v.type;
    o.random_phase = v.random_phase;
    o.omega_dt = v.omega_dt;
#pragma unroll
    for (int i = 0; i < 3; ++i) { o.amp[i] = (float)v.amp[i]; o.phase[i] = 0.0; }
    if (v.type == GINSIM_VIB_SINUSOIDAL && v.random_phase) {
        const u32x4 w = philox4x32(0u, PHASE_STREAM >> 1, key.r0, key.r1, key.k0, key.k1);
        o.phase[0] = ((double)w.x * 0x1p-32 * 2.0) * 3.14159265358979323846;
        o.phase[1] = ((double)w.y * 0x1p-32 * 2.0) * 3.14159265358979323846;
        o.phase[2] = ((double)w.z * 0x1p-32 * 2.0) * 3.14159265358979323846;
    }
}

template <uint32_t STREAM>
F32_FM V3 add_vibration(const V3& o, const Vib& v, const RngKey& key, uint32_t j, const NormalTables& tab) {
    V3 r = o;
    if (v.type == GINSIM_VIB_RANDOM) {
        float z0[2], z1[2];
        normal_pairs_f32<STREAM, 2>(key, j, z0, z1, tab);
        r.x = fm(v.amp[0], z0[0], o.x);
        r.y = fm(v.amp[1], z1[0], o.y);
        r.z = fm(v.amp[2], z0[1], o.z);
    } else if (v.type == GINSIM_VIB_SINUSOIDAL) {
        const double cj = v.omega_dt * (double)j;
        float sn[3], cs;
        if (v.random_phase) {
            sincos_def(cj + v.phase[0], sn[0], cs);
            sincos_def(cj + v.phase[1], sn[1], cs);
            sincos_def(cj + v.phase[2], sn[2], cs);
        } else {
            sincos_def(cj, sn[0], cs);
            sn[1] = sn[0];
            sn[2] = sn[0];
        }
        r.x = fm(v.amp[0], sn[0], o.x);
        r.y = fm(v.amp[1], sn[1], o.y);
        r.z = fm(v.amp[2], sn[2], o.z);
    }
    return r;
}

// first normal of the odometer stream (S_ODO = 6: the low half of block 3) at sample j
F32_FM float odo_normal(const RngKey& key, uint32_t j, const NormalTables& tab) {
    const u32x4 w = philox4x32(j, S_ODO >> 1, key.r0, key.r1, key.k0, key.k1);
    const uint32_t a[1] = {w.x}, b[1] = {w.y};
    float z0[1], z1[1];
    normal_transform<1>(a, b, z0, z1, tab);
    return z0[0];
}

template <int RF, int ALGOS, bool GIVEN, bool WD, bool VIB = false>
__global__ void __launch_bounds__(256, 2) mc_kernel_f32(const ginsim_mc_params a) {
    static_assert(!VIB || (!GIVEN && WD), "vibration: generate mode, general sensor model");
    __shared__ uint32_t ntab[GIVEN ? 4 : kNormalLdsWords];
    NormalTables tab{};
    if (!GIVEN) {
        tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
        __syncthreads();
    }
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.runs) return;
    constexpr bool FREE = (ALGOS & GINSIM_ALGO_FREE) != 0;
    constexpr bool ODO = (ALGOS & GINSIM_ALGO_ODO) != 0;
    const int64_t n = a.n, runs = a.runs, plane = n * runs;
    const float dt = (float)(1.0 / a.fs);
    const uint64_t call = a.ini_first + (uint64_t)r;
    const double* ini = a.ini + 10 * (call < (uint64_t)a.n_ini ? call : 0);
    Nav fi, od;
    if (FREE) nav_init<RF>(fi, ini, a.ini_has_g);
    if (ODO) nav_init<RF>(od, ini, a.ini_has_g);
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    Model ma, mg;
    load_model(a.accel, ma);
    load_model(a.gyro, mg);
    const float odo_scale = (float)a.odo_scale, odo_stdv = (float)a.odo_stdv;
    float da[3] = {0.f, 0.f, 0.f}, dg[3] = {0.f, 0.f, 0.f};
    Vib va, vg;
    if (VIB) {
        load_vib<S_ACC_VIB_PHASE>(a.vib_accel, key, va);
        load_vib<S_GYR_VIB_PHASE>(a.vib_gyro, key, vg);
    }
    float* o_acc = reinterpret_cast<float*>(a.out_accel);
    float* o_gyr = reinterpret_cast<float*>(a.out_gyro);
    float* o_odo = reinterpret_cast<float*>(a.out_odo);
    float* o_fi = reinterpret_cast<float*>(a.out_traj[0]);
    float* o_od = reinterpret_cast<float*>(a.out_traj[1]);
    const uniform_ptr ref_a = (uniform_ptr)(uintptr_t)a.ref_accel, ref_g = (uniform_ptr)(uintptr_t)a.ref_gyro,
                      ref_o = (uniform_ptr)(uintptr_t)a.ref_odo;
    if (FREE && o_fi) store9<RF>(o_fi, plane, r, fi);
    if (ODO && o_od) store9<RF>(o_od, plane, r, od);
    for (int64_t j = 0; j < n; ++j) {
        const int64_t off = j * runs + r;
        V3 acc{0.f, 0.f, 0.f}, gyr;
        float odo = 0.f;
        if (GIVEN) {    // the fp64 series of the plugin boundary, rounded to float as they are read
            if (j == n - 1) break;
            gyr = V3{(float)a.in_gyro[off], (float)a.in_gyro[plane + off], (float)a.in_gyro[2 * plane + off]};
            if (FREE) acc = V3{(float)a.in_accel[off], (float)a.in_accel[plane + off], (float)a.in_accel[2 * plane + off]};
            if (ODO) odo = (float)a.in_odo[off];
        } else {
            const bool last = (j == n - 1);
            if (last && !o_acc && !o_gyr && !o_odo) break;
            const uint32_t jj = (uint32_t)j;
            // wave-uniform truth of this step, requested before the noise is generated (scalar-load latency hidden)
            const double ta[3] = {ref_a[3 * j], ref_a[3 * j + 1], ref_a[3 * j + 2]};
            const double tg[3] = {ref_g[3 * j], ref_g[3 * j + 1], ref_g[3 * j + 2]};
            // the twelve normals of the step: streams 0..5 = three Philox blocks (z0 / z1 of stream s; philox.hpp)
            float z0[6], z1[6];
            normal_pairs_f32<S_ACC_D_XY, 6>(key, jj, z0, z1, tab);
            const float zda[3] = {z0[0], z1[0], z0[1]}, zwa[3] = {z1[1], z0[2], z1[2]};
            const float zdg[3] = {z0[3], z1[3], z0[4]}, zwg[3] = {z1[4], z0[5], z1[5]};
            acc = sense3<WD>(ta, ma, da, zda, zwa);
            gyr = sense3<WD>(tg, mg, dg, zdg, zwg);
            if (VIB) {
                acc = add_vibration<S_ACC_VIB_XY>(acc, va, key, jj, tab);
                gyr = add_vibration<S_GYR_VIB_XY>(gyr, vg, key, jj, tab);
            }
            if (o_acc) { st(o_acc + off, acc.x); st(o_acc + plane + off, acc.y); st(o_acc + 2 * plane + off, acc.z); }
            if (o_gyr) { st(o_gyr + off, gyr.x); st(o_gyr + plane + off, gyr.y); st(o_gyr + 2 * plane + off, gyr.z); }
            if (ODO || o_odo) {
                odo = fm(odo_stdv, odo_normal(key, jj, tab), odo_scale * (float)ref_o[j]);     // pathgen.py:639-640
                if (o_odo) st(o_odo + off, odo);
            }
            if (last) break;
        }
        const bool resync = ((j + 1) & (kTrigResync - 1)) == 0;
        if (FREE) {
            nav_step<RF, false>(fi, gyr, acc, 0.f, dt, a.earth_rot, resync);
            if (o_fi) store9<RF>(o_fi, plane, off + runs, fi);
        }
        if (ODO) {
            nav_step<RF, true>(od, gyr, acc, odo, dt, a.earth_rot, resync);
            if (o_od) store9<RF>(o_od, plane, off + runs, od);
        }
    }
    if (FREE && a.out_end[0]) store_end<RF>(a.out_end[0], runs, r, fi);
    if (ODO && a.out_end[1]) store_end<RF>(a.out_end[1], runs, r, od);
}

// Wave-specialised variant (see mc_kernel_split in mc_kernel.hip for the rationale): the producer wavefronts of a
// workgroup generate the twelve normals of a step -- the SAME code as the fp64 kernel's producers -- into an LDS ring,
// waves 0-3 consume them and do sensors, mechanisation and stores.  PROD producer wavefronts per consumer wavefront
// (the steps of a tile alternate between the groups).  Bit-identical to mc_kernel_f32.
//
// The consumer is ONE wavefront per SIMD and the step time hangs on it: a lone wavefront issues an instruction every ~5
// cycles whatever its kind (tools/experiments/ubench3.hip: v_fma_f32 5.1, s_mov_b32 8.5, v_cvt_f32_f64 8.0, v_mad_u64_u32 9.1, a
// transcendental 8.8), so its loop is written for instruction COUNT:
//   * KEEP is a template parameter (everything kept / nothing kept; other combinations take the plain kernel): no
//     per-step tests of the output pointers;
//   * the series are stored through buffer resources -- one descriptor per group of three planes, ONE 32-bit lane offset
//     advanced by runs x 4 bytes per step, the plane as the scalar offset: 15 x `buffer_store_dword v, voff, rsrc, splane`
//     and one v_add_u32 per step instead of 15 64-bit address computations (needs 3 planes < 4 GiB: the launcher checks);
//   * the truth of a step comes as eight floats {accel xyz, gyro xyz, odometer, 0} converted once per launch by
//     truth_f32_kernel (one s_load_dwordx8, no six v_cvt_f32_f64 per step);
//   * the rare work of the attitude update sits behind one wave-uniform branch (Att::step).
constexpr int kSplitTileF = 6;
constexpr int kSplitRunsF = 256;
constexpr size_t kSplitLdsF = sizeof(float) * 2 * kSplitTileF * 12 * kSplitRunsF;      // 144 KiB

typedef const float __attribute__((address_space(4))) * uniform_f32_ptr;

struct Planes3 {            // three consecutive [n][runs] float planes behind one buffer descriptor
    __amdgpu_buffer_rsrc_t rs;
    F32_FM void init(float* base) { rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1 /* 4 GiB - 1 */, 0x00020000); }
    F32_FM void store(uint32_t voff, uint32_t pl, float x, float y, float z) const {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, voff, 0, 2);         // aux 2 = nt
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs, voff, pl, 2);
        // the doubled offset is made a scalar here: left to the compiler it ended up in a VGPR and every third store in a
        // nine-instruction waterfall loop (v_readfirstlane / v_cmp / s_and_saveexec ...), five of them per step
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z), rs, voff, __builtin_amdgcn_readfirstlane(2 * pl), 2);
    }
};

struct TrajOut {
    Planes3 att, pos, vel;
    F32_FM void init(float* base, int64_t plane) { att.init(base); pos.init(base + 3 * plane); vel.init(base + 6 * plane); }
    template <int RF>
    F32_FM void store(uint32_t voff, uint32_t pl, const Nav& s) const {
        att.store(voff, pl, s.att.yaw.v, s.att.pit.v, s.att.rol.v);
        if (RF == 1) pos.store(voff, pl, (float)s.pos[0], (float)s.pos[1], (float)s.pos[2]);
        else pos.store(voff, pl, (float)(s.pos[0] - s.pos0[0]), (float)(s.pos[1] - s.pos0[1]), (float)(s.pos[2] - s.pos0[2]));
        vel.store(voff, pl, s.vel.x, s.vel.y, s.vel.z);
    }
};

// truth [n][3] + [n][3] (+ [n]) doubles -> [n][8] floats {accel xyz, gyro xyz, odo, 0}: the conversions the sensor sums would
// otherwise repeat in every lane's instruction stream (same rounding: one v_cvt_f32_f64 each)
__global__ void truth_f32_kernel(const double* __restrict__ ra, const double* __restrict__ rg, const double* __restrict__ ro,
                                 int64_t n, float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float* o = out + 8 * j;
    o[0] = (float)ra[3 * j]; o[1] = (float)ra[3 * j + 1]; o[2] = (float)ra[3 * j + 2];
    o[3] = (float)rg[3 * j]; o[4] = (float)rg[3 * j + 1]; o[5] = (float)rg[3 * j + 2];
    o[6] = ro ? (float)ro[j] : 0.f;
    o[7] = 0.f;
}

template <int RF, int ALGOS, bool WD, int PROD, bool KEEP>
__global__ void __launch_bounds__(256 * (1 + PROD)) mc_kernel_f32_split(const ginsim_mc_params a, const float* __restrict__ truth32) {
    extern __shared__ float zringf[];                   // [2 stages][T steps][12 normals][256 runs]
    constexpr bool FREE = (ALGOS & GINSIM_ALGO_FREE) != 0;
    constexpr bool ODO = (ALGOS & GINSIM_ALGO_ODO) != 0;
    constexpr int kStepFloats = 12 * kSplitRunsF;
    const int lane = threadIdx.x & (kSplitRunsF - 1);
    const bool producer = threadIdx.x >= kSplitRunsF;
    const int pgroup = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) - 1;     // which producer group (wave-uniform)
    const int64_t r = (int64_t)blockIdx.x * kSplitRunsF + lane;
    const bool active = r < a.runs;
    const uint32_t n = (uint32_t)a.n;
    const int64_t runs = a.runs;
    const uint32_t n_noise = KEEP ? n : n - 1;            // the last sample only exists as sensor output
    const uint32_t ntiles = (n_noise + kSplitTileF - 1) / kSplitTileF;
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();
#ifdef GINSIM_EXPERIMENT
    const int exp_flags = a.accel.reserved;
#else
    constexpr int exp_flags = 0;
#endif

    if (producer) {
        for (uint32_t i = 0; i <= ntiles; ++i) {
            if (i < ntiles && active && !(exp_flags & 1)) {
                float* stage = zringf + (i & 1) * (kSplitTileF * kStepFloats);
#pragma unroll
                for (int t = 0; t < kSplitTileF; ++t) {
                    const uint32_t j = i * kSplitTileF + t;
                    if (j < n_noise && (PROD == 1 || (t % PROD) == pgroup)) {
                        float z0[6], z1[6];
                        normal_pairs_f32<S_ACC_D_XY, 6>(key, j, z0, z1, tab);
                        float* zb = stage + t * kStepFloats + lane;
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            zb[(2 * k) * kSplitRunsF] = z0[k];
                            zb[(2 * k + 1) * kSplitRunsF] = z1[k];
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    const float dt = (float)(1.0 / a.fs);
    const uint64_t call = a.ini_first + (uint64_t)r;
    const double* ini = a.ini + 10 * ((active && call < (uint64_t)a.n_ini) ? call : 0);
    Nav fi, od;
    if (FREE) nav_init<RF>(fi, ini, a.ini_has_g);
    if (ODO) nav_init<RF>(od, ini, a.ini_has_g);
    Model ma, mg;
    load_model(a.accel, ma);
    load_model(a.gyro, mg);
    const float odo_scale = (float)a.odo_scale, odo_stdv = (float)a.odo_stdv;
    float da[3] = {0.f, 0.f, 0.f}, dg[3] = {0.f, 0.f, 0.f};
    const uniform_f32_ptr truth = (uniform_f32_ptr)(uintptr_t)truth32;
    Planes3 o_acc, o_gyr, o_odo;
    TrajOut o_fi, o_od;
    const int64_t plane = (int64_t)n * runs;
    const uint32_t pl = __builtin_amdgcn_readfirstlane((uint32_t)plane * 4u), step_bytes = (uint32_t)runs * 4u;
    uint32_t voff = (uint32_t)r * 4u;                   // byte offset of (sample j, run r) inside a plane
    if (KEEP) {
        o_acc.init(reinterpret_cast<float*>(a.out_accel));
        o_gyr.init(reinterpret_cast<float*>(a.out_gyro));
        if (ODO) o_odo.init(reinterpret_cast<float*>(a.out_odo));
        if (FREE) o_fi.init(reinterpret_cast<float*>(a.out_traj[0]), plane);
        if (ODO) o_od.init(reinterpret_cast<float*>(a.out_traj[1]), plane);
        if (active) {
            if (FREE) o_fi.store<RF>(voff, pl, fi);
            if (ODO) o_od.store<RF>(voff, pl, od);
        }
    }
    // sensor outputs of sample j (kept or not) from the ring's normals at `zb`
    auto sensors = [&](uint32_t j, const float* zb, V3& acc, V3& gyr, float& odo) {
        const uniform_f32_ptr tj = truth + 8 * (uint64_t)j;
        const double ta[3] = {(double)tj[0], (double)tj[1], (double)tj[2]};      // widened and narrowed again: exact
        const double tg[3] = {(double)tj[3], (double)tj[4], (double)tj[5]};
        float p0[6], p1[6];                   // z0 / z1 of streams 0..5
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            p0[k] = zb[(2 * k) * kSplitRunsF];
            p1[k] = zb[(2 * k + 1) * kSplitRunsF];
        }
        const float zda[3] = {p0[0], p1[0], p0[1]}, zwa[3] = {p1[1], p0[2], p1[2]};
        const float zdg[3] = {p0[3], p1[3], p0[4]}, zwg[3] = {p1[4], p0[5], p1[5]};
        acc = sense3<WD>(ta, ma, da, zda, zwa);
        gyr = sense3<WD>(tg, mg, dg, zdg, zwg);
        if (KEEP) {
            o_acc.store(voff, pl, acc.x, acc.y, acc.z);
            o_gyr.store(voff, pl, gyr.x, gyr.y, gyr.z);
        }
        odo = 0.f;
        if (ODO) {
            odo = fm(odo_stdv, odo_normal(key, j, tab), odo_scale * tj[6]);
            if (KEEP) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(odo), o_odo.rs, voff, 0, 2);
        }
    };
    // samples 0 .. n-2 have a mechanisation step behind them: a COUNTED inner loop without a way out in the middle of its
    // body (the old `if (j == n - 1) break` after the sensors gave the loop a second merge block fed by the un-stepped
    // state).  The last sample, sensor output only, follows the loops: its normals are still in the ring.
    const uint32_t n_steps = n - 1;
    for (uint32_t i = 0; i <= ntiles; ++i) {
        if (i >= 1 && active && !(exp_flags & 2)) {
            const float* stage = zringf + ((i - 1) & 1) * (kSplitTileF * kStepFloats);
            const uint32_t j0 = (i - 1) * kSplitTileF;
            const uint32_t left = j0 < n_steps ? n_steps - j0 : 0u;
            const int cnt = (int)(left < (uint32_t)kSplitTileF ? left : (uint32_t)kSplitTileF);
            // two steps per trip: the loop-carried state of the second step lands where the first step of the next trip expects
            // it, which the allocator did not manage with one step per trip (sixteen v_mov at the back edge)
            auto one_step = [&](int t) {
                const uint32_t j = j0 + t;
                V3 acc, gyr;
                float odo;
                sensors(j, stage + t * kStepFloats + lane, acc, gyr, odo);
                voff += step_bytes;
                const bool resync = ((j + 1) & (kTrigResync - 1)) == 0;
                if (FREE) {
                    nav_step<RF, false>(fi, gyr, acc, 0.f, dt, a.earth_rot, resync);
                    if (KEEP) o_fi.store<RF>(voff, pl, fi);
                }
                if (ODO) {
                    nav_step<RF, true>(od, gyr, acc, odo, dt, a.earth_rot, resync);
                    if (KEEP) o_od.store<RF>(voff, pl, od);
                }
            };
            int t = 0;
            if constexpr (!(FREE && ODO)) {         // with both algorithms the doubled body would spill
#pragma unroll 1
                for (; t + 1 < cnt; t += 2) {
                    one_step(t);
                    one_step(t + 1);
                }
                if (t < cnt) one_step(t);
            } else {
#pragma unroll 1
                for (; t < cnt; ++t) one_step(t);
            }
        }
        __syncthreads();
    }
    if (KEEP && active && !(exp_flags & 2)) {
        V3 acc, gyr;
        float odo;
        sensors(n_steps, zringf + ((n_steps / kSplitTileF) & 1) * (kSplitTileF * kStepFloats) + (n_steps % kSplitTileF) * kStepFloats + lane, acc, gyr, odo);
    }
    if (active) {
        if (FREE && a.out_end[0]) store_end<RF>(a.out_end[0], runs, r, fi);
        if (ODO && a.out_end[1]) store_end<RF>(a.out_end[1], runs, r, od);
    }
}

}  // namespace f32

static int split_policy_f32() {
    static const int v = [] { const char* e = getenv("GINSIM_SPLIT"); return e ? atoi(e) : -1; }();
    return v;
}

// producer groups of the wave-specialised fp32 kernel asked for by GINSIM_SPLIT_PROD (1..3), 0: the default of the variant
static int split_prod_f32() {
    static const int v = [] { const char* e = getenv("GINSIM_SPLIT_PROD"); const int k = e ? atoi(e) : 0; return k >= 1 && k <= 3 ? k : 0; }();
    return v;
}

// everything this launch can keep is kept (1) / nothing is (0) / a mixture (-1: the plain kernel tests the pointers per step)
static int keep_mode_f32(const ginsim_mc_params& p) {
    const bool fre = (p.algo_mask & GINSIM_ALGO_FREE) != 0, odo = (p.algo_mask & GINSIM_ALGO_ODO) != 0;
    const bool all = p.out_accel && p.out_gyro && (!fre || p.out_traj[0]) && (!odo || (p.out_traj[1] && p.out_odo)) && (odo || !p.out_odo);
    const bool none = !p.out_accel && !p.out_gyro && !p.out_odo && !p.out_traj[0] && !p.out_traj[1];
    return all ? 1 : (none ? 0 : -1);
}

static bool any_vibration_f32(const ginsim_mc_params& p) { return p.vib_accel.type != GINSIM_VIB_NONE || p.vib_gyro.type != GINSIM_VIB_NONE; }

int mc_variant_f32(const ginsim_mc_params& p) {
    if (any_vibration_f32(p)) return 0;         // the vibration term lives in the plain kernel (general sensor model)
    if (!(p.algo_mask & GINSIM_ALGO_FREE) || p.given_sensors || p.block_threads != 0 || p.n < 2) return 0;      // block_threads: the plain kernel (tests)
    const int keep = keep_mode_f32(p);
    if (keep < 0) return 0;
    // the wave-specialised kernel addresses a group of three planes with 32-bit byte offsets
    if (keep == 1 && (double)p.n * (double)p.runs * 12.0 >= 4294967296.0) return 0;
    const int pol = split_policy_f32();
    if (pol >= 0) return pol != 0;
    return 1;       // three wavefronts per SIMD beat the plain kernel at every size (as for the fp64 kernel)
}

static bool any_white_drift_f32(const ginsim_mc_params& p) {
    bool f = false;
    for (int k = 0; k < 3; ++k)
        f = f || p.accel.white_drift[k] || p.gyro.white_drift[k] || p.accel.bias[k] != 0.0 || p.gyro.bias[k] != 0.0;
    return f;
}

template <int RF, int ALGOS, bool WD, int PROD, bool KEEP>
static hipError_t launch_split_f32(const ginsim_mc_params& p, const float* truth32, hipStream_t stream) {
    static PerDeviceOnce once;
    once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&f32::mc_kernel_f32_split<RF, ALGOS, WD, PROD, KEEP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)f32::kSplitLdsF);
    });
    hipLaunchKernelGGL((f32::mc_kernel_f32_split<RF, ALGOS, WD, PROD, KEEP>), dim3((unsigned)((p.runs + 255) / 256)), dim3(256 * (1 + PROD)),
                       f32::kSplitLdsF, stream, p, truth32);
    return hipGetLastError();
}

// name != nullptr: write the kernel's name (as rocprofv3 reports it, without arguments) instead of launching
template <int RF, int ALGOS, bool WD>
static hipError_t launch3_f32(const ginsim_mc_params& p, const float* truth32, hipStream_t stream, char* name, size_t cap) {
    const int tb = 256;
    const int64_t waves = (p.runs + 63) / 64;
    if constexpr ((ALGOS & GINSIM_ALGO_FREE) != 0) {
        if (mc_variant_f32(p)) {
            // producer groups: three (four wavefronts per SIMD, <= 128 registers) with one algorithm, two (<= 168) with both --
            // one group fewer with the per-sample drift model, which would spill at that bound
            constexpr int MAXP = ALGOS == GINSIM_ALGO_FREE ? (WD ? 2 : 3) : (WD ? 1 : 2);
            const int want = split_prod_f32();
            const int prod = want == 0 || want > MAXP ? MAXP : want;
            const bool keep = keep_mode_f32(p) == 1;
            if (name) {
                snprintf(name, cap, "ginsim::f32::mc_kernel_f32_split<%d, %d, %s, %d, %s>", RF, ALGOS, WD ? "true" : "false", prod,
                         keep ? "true" : "false");
                return hipSuccess;
            }
            hipLaunchKernelGGL(f32::truth_f32_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, stream, p.ref_accel, p.ref_gyro,
                               p.ref_odo, p.n, const_cast<float*>(truth32));
            if constexpr (MAXP == 3) {
                if (prod == 3)
                    return keep ? launch_split_f32<RF, ALGOS, WD, 3, true>(p, truth32, stream) : launch_split_f32<RF, ALGOS, WD, 3, false>(p, truth32, stream);
            }
            if constexpr (MAXP >= 2) {
                if (prod == 2)
                    return keep ? launch_split_f32<RF, ALGOS, WD, 2, true>(p, truth32, stream) : launch_split_f32<RF, ALGOS, WD, 2, false>(p, truth32, stream);
            }
            return keep ? launch_split_f32<RF, ALGOS, WD, 1, true>(p, truth32, stream) : launch_split_f32<RF, ALGOS, WD, 1, false>(p, truth32, stream);
        }
    }
    const bool vib = WD && any_vibration_f32(p);
    if (name) {
        snprintf(name, cap, "ginsim::f32::mc_kernel_f32<%d, %d, false, %s, %s>", RF, ALGOS, WD ? "true" : "false", vib ? "true" : "false");
        return hipSuccess;
    }
    // exactly k workgroups per CU (k + 1 do not fit the LDS reservation)
    const int per_cu = waves <= 1024 ? 1 : (waves <= 2048 ? 2 : 3);
    const size_t lds = (160 * 1024) / (per_cu + 1) + 1024 - sizeof(uint32_t) * kNormalLdsWords;
    if constexpr (WD) {
        if (vib) {
            hipLaunchKernelGGL((f32::mc_kernel_f32<RF, ALGOS, false, true, true>), dim3((unsigned)((p.runs + tb - 1) / tb)), dim3(tb), lds, stream, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((f32::mc_kernel_f32<RF, ALGOS, false, WD>), dim3((unsigned)((p.runs + tb - 1) / tb)), dim3(tb), lds, stream, p);
    return hipGetLastError();
}

template <int RF, int ALGOS>
static hipError_t launch2_f32(const ginsim_mc_params& p, const float* truth32, hipStream_t stream, char* name, size_t cap) {
    if (p.given_sensors) {
        if (name) {
            snprintf(name, cap, "ginsim::f32::mc_kernel_f32<%d, %d, true, false, false>", RF, ALGOS);
            return hipSuccess;
        }
        const int64_t waves = (p.runs + 63) / 64;
        const int per_cu = waves <= 1024 ? 1 : (waves <= 2048 ? 2 : 3);
        const size_t lds = (160 * 1024) / (per_cu + 1) + 1024;
        hipLaunchKernelGGL((f32::mc_kernel_f32<RF, ALGOS, true, false>), dim3((unsigned)((p.runs + 255) / 256)), dim3(256), lds, stream, p);
        return hipGetLastError();
    }
    return any_white_drift_f32(p) || any_vibration_f32(p) ? launch3_f32<RF, ALGOS, true>(p, truth32, stream, name, cap)
                                                          : launch3_f32<RF, ALGOS, false>(p, truth32, stream, name, cap);
}

template <int RF>
static hipError_t launch1_f32(const ginsim_mc_params& p, const float* truth32, hipStream_t stream, char* name, size_t cap) {
    switch (p.algo_mask) {
        case GINSIM_ALGO_FREE: return launch2_f32<RF, GINSIM_ALGO_FREE>(p, truth32, stream, name, cap);
        case GINSIM_ALGO_ODO: return launch2_f32<RF, GINSIM_ALGO_ODO>(p, truth32, stream, name, cap);
        default: return launch2_f32<RF, GINSIM_ALGO_FREE | GINSIM_ALGO_ODO>(p, truth32, stream, name, cap);
    }
}

// truth32: device scratch of n x 8 floats (mc_f32_truth_bytes) the wave-specialised kernel's truth is converted into
size_t mc_f32_truth_bytes(const ginsim_mc_params& p) { return mc_variant_f32(p) ? sizeof(float) * 8 * (size_t)p.n : 0; }

hipError_t launch_mc_f32(const ginsim_mc_params& p_in, float* truth32, hipStream_t stream, char* name, size_t cap) {
    ginsim_mc_params p = p_in;
#ifdef GINSIM_EXPERIMENT
    static const int exp = [] { const char* e = getenv("GINSIM_EXP"); return e ? atoi(e) : 0; }();
    p.accel.reserved = exp;
#endif
    return p.ref_frame == 1 ? launch1_f32<1>(p, truth32, stream, name, cap) : launch1_f32<0>(p, truth32, stream, name, cap);
}

// gather selected runs of a float series: [C][n][runs] (float) -> out [nsel][n][C] (double), optional per-component origin
__global__ void gather_runs_f32_kernel(const float* __restrict__ series, int C, int64_t n, int64_t runs,
                                       const int64_t* __restrict__ ids, int nsel, double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)nsel * n * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t j = (idx / C) % n;
    const int64_t k = idx / (C * n);
    out[idx] = (double)series[((int64_t)c * n + j) * runs + ids[k]];
}

hipError_t launch_gather_runs_f32(const float* series, int C, int64_t n, int64_t runs, const int64_t* ids, int nsel,
                                  double* out, hipStream_t s) {
    const int tb = 256;
    const int64_t total = (int64_t)nsel * n * C;
    hipLaunchKernelGGL(gather_runs_f32_kernel, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, s, series, C, n, runs,
                       ids, nsel, out);
    return hipGetLastError();
}

}  // namespace ginsim
