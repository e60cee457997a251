// Fused Monte-Carlo kernel: sensor-error injection + strapdown mechanisation + end-point error.
//
// One lane = one Monte-Carlo run (no inter-lane traffic in the time loop; workgroups of 256 or 512 threads only
// shape the placement on the SIMDs and share the coefficient table (8 KB of LDS) of the normal transform).  Per-run state (Euler attitude + cached
// trig, body/NED velocity, position, the six Gauss-Markov bias states) lives in VGPRs for the whole time loop.
// Truth samples are wave-uniform and come in through the scalar cache.  Everything that leaves the lane is SoA [component][sample][run]
// (run fastest) so that every store instruction of a wavefront writes 64 x 8 B contiguous bytes.
//
// Restates, per run:
//   Sim.__gen_data_from_pathgen loop body      gnss_ins_sim/sim/ins_sim.py:490-506
//   pathgen.acc_gen / gyro_gen / bias_drift    gnss_ins_sim/pathgen/pathgen.py:441-594
//   pathgen.odo_gen                            gnss_ins_sim/pathgen/pathgen.py:627-641
//   FreeIntegration.run                        demo_algorithms/free_integration.py:63-174
//   FreeIntegration.run (odometer variant)     demo_algorithms/free_integration_odo.py:63-160
//   array_error + end-point pick               gnss_ins_sim/sim/ins_data_manager.py:519-541, 737
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "ginsim.h"
#include "ins_math.hpp"
#include "philox.hpp"
#include "device_once.hpp"

namespace ginsim {

constexpr int kWave = 64;

// The by-value parameter block sits at offset 0 of the kernarg segment.  Its bulky members (two sensor models,
// ref_end) are re-read from there with scalar loads where they are used: kept in SGPRs for the whole time loop
// they overflow the SGPR file and the compiler parks them in VGPR lanes (v_readlane/v_writelane = VALU issue
// slots, ~14 % of the loop in the first build).  The empty asm makes the pointer opaque per iteration so the
// loads are not hoisted back out of the loop.
typedef const ginsim_mc_params __attribute__((address_space(4))) * params_ptr;
__device__ __forceinline__ params_ptr kernarg_params() {
    params_ptr p = (params_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
typedef const ginsim_sensor_model __attribute__((address_space(4))) * model_ptr;


// One strapdown solution (one algorithm instance of one run).
struct Nav {
    Att  att;
    Vec3 vb;    // body velocity (ref_frame 1, free integration)
    Vec3 vel;   // navigation-frame velocity of the previous sample
    Vec3 pos;   // ECEF+displacement (ref_frame 1) or LLA (ref_frame 0)
    double g;   // gravity: constant (ref_frame 1) or the external override of ref_frame 0
    double sl, cl;  // ref_frame 0: cached sin/cos of the latitude pos.x
    bool ext_g; // ref_frame 0: use g instead of the WGS-84 model (free_integration.py:143-146)
};

template <int RF>
__device__ __forceinline__ void nav_init(Nav& s, const double* __restrict__ ini, int has_g) {
    // free_integration.py:96-102 / :127-131
    s.att.set(ini[6], ini[7], ini[8]);
    s.vb = Vec3{ini[3], ini[4], ini[5]};
    s.vel = s.att.to_nav(s.vb);
    if (RF == 1) {
        s.pos = lla2ecef(ini[0], ini[1], ini[2]);
        s.g = has_g ? ini[9] : geo_param(ini[0], ini[2]).g;     // free_integration.py:89-93
    } else {
        s.pos = Vec3{ini[0], ini[1], ini[2]};
        s.g = has_g ? ini[9] : 0.0;
        sincos(ini[0], &s.sl, &s.cl);
    }
    s.ext_g = has_g != 0;
}

// One time step.  ODO == false: free_integration.py:104-116 (RF 1) / :134-172 (RF 0);
//                 ODO == true : free_integration_odo.py:96-105 (RF 1) / :118-152 (RF 0).
template <int RF, bool ODO, bool EASY = false>
__device__ __forceinline__ void nav_step(Nav& s, const Vec3& gyro, const Vec3& accel, double odo, double dt,
                                         int earth_rot, bool resync, const MathConsts& mk) {
    if (RF == 1) {
        const Vec3 v_prev = s.vel;
        if (!ODO) {
            const Vec3 gb = s.att.down_in_body();          // C(att[i-1]) . [0,0,g]
            const Vec3 wxv = cross3(gyro, s.vb);
            s.vb.x += (accel.x + gb.x * s.g) * dt - wxv.x * dt;
            s.vb.y += (accel.y + gb.y * s.g) * dt - wxv.y * dt;
            s.vb.z += (accel.z + gb.z * s.g) * dt - wxv.z * dt;
        }
        s.att.template step<EASY>(gyro, dt, resync, mk);
        if (ODO) {
            const Vec3 f = s.att.fwd_in_nav();
            s.vel = Vec3{f.x * odo, f.y * odo, f.z * odo};
        } else {
            s.vel = s.att.to_nav(s.vb);
        }
        s.pos.x += v_prev.x * dt;
        s.pos.y += v_prev.y * dt;
        s.pos.z += v_prev.z * dt;
    } else {
        const Geo e = geo_param_sc(s.sl, s.cl, s.pos.z);
        const double irm = rcp_n1(e.rm + s.pos.z);     // one Newton step (2^-46): these scale rates of ~1e-6 rad/s
        const double irn = rcp_n1(e.rn + s.pos.z);
        const double icl = rcp_n1(e.cl);
        const Vec3 v = s.vel;
        const Vec3 w_en{v.y * irn, -v.x * irm, -v.y * e.sl * icl * irn};
        Vec3 w_ie{0.0, 0.0, 0.0};
        if (earth_rot) { w_ie.x = kWie * e.cl; w_ie.z = -kWie * e.sl; }
        const Vec3 wb = s.att.to_body(Vec3{w_en.x + w_ie.x, w_en.y + w_ie.y, w_en.z + w_ie.z});
        const Vec3 w_nb{gyro.x - wb.x, gyro.y - wb.y, gyro.z - wb.z};
        Vec3 v_new;
        if (!ODO) {
            const Vec3 an = s.att.to_nav(accel);            // C(att[i-1])^T accel
            const double g = s.ext_g ? s.g : e.g;
            const Vec3 cor = cross3(Vec3{2.0 * w_ie.x + w_en.x, 2.0 * w_ie.y + w_en.y, 2.0 * w_ie.z + w_en.z}, v);
            v_new = Vec3{v.x + (an.x - cor.x) * dt, v.y + (an.y - cor.y) * dt, v.z + (an.z + g - cor.z) * dt};
        }
        s.att.template step<EASY>(w_nb, dt, resync, mk);
        if (ODO) {
            const Vec3 f = s.att.fwd_in_nav();
            v_new = Vec3{f.x * odo, f.y * odo, f.z * odo};
        }
        const double dlat = v.x * irm * dt;
        if (EASY && __builtin_amdgcn_ballot_w64(resync || !(fabs(dlat) <= 0x1.0p-6)) == 0) rotate_sincos_small(dlat, s.sl, s.cl, mk);
        else if (resync || !(fabs(dlat) <= 0.25)) sincos(s.pos.x + dlat, &s.sl, &s.cl);
        else if (fabs(dlat) <= 0x1.0p-6) rotate_sincos_small(dlat, s.sl, s.cl, mk);      // per lane: see Att::step
        else rotate_sincos(dlat, s.sl, s.cl, mk);
        s.pos.x += dlat;
        s.pos.y += v.y * irn * icl * dt;
        s.pos.z += -v.z * dt;
        s.vel = v_new;
    }
}

// Series are written once and not read back by the kernel: non-temporal stores keep them from displacing the L2.
__device__ __forceinline__ void st(double* p, double v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ void store9(double* __restrict__ base, int64_t plane, int64_t off, const Nav& s) {
    st(base + 0 * plane + off, s.att.yaw);
    st(base + 1 * plane + off, s.att.pit);
    st(base + 2 * plane + off, s.att.rol);
    st(base + 3 * plane + off, s.pos.x);
    st(base + 4 * plane + off, s.pos.y);
    st(base + 5 * plane + off, s.pos.z);
    st(base + 6 * plane + off, s.vel.x);
    st(base + 7 * plane + off, s.vel.y);
    st(base + 8 * plane + off, s.vel.z);
}

__device__ __forceinline__ void store3(double* __restrict__ base, int64_t plane, int64_t off, const Vec3& v) {
    st(base + off, v.x);
    st(base + plane + off, v.y);
    st(base + 2 * plane + off, v.z);
}

// the second end-point record of ref_frame 0 launches: same attitude / velocity errors, position error in NED metres
__device__ __forceinline__ void store_end_ned(double* __restrict__ out, int64_t runs, int64_t r, const Nav& s) {
    const params_ptr kp = kernarg_params();
    const double ref_end[9] = {kp->ref_end[0], kp->ref_end[1], kp->ref_end[2], kp->ref_end[3], kp->ref_end[4],
                               kp->ref_end[5], kp->ref_end[6], kp->ref_end[7], kp->ref_end[8]};
    out[0 * runs + r] = angle_range_pi(s.att.yaw - ref_end[0]);
    out[1 * runs + r] = angle_range_pi(s.att.pit - ref_end[1]);
    out[2 * runs + r] = angle_range_pi(s.att.rol - ref_end[2]);
    const Vec3 ep = lla_error_ned(s.pos, Vec3{ref_end[3], ref_end[4], ref_end[5]});
    out[3 * runs + r] = ep.x;
    out[4 * runs + r] = ep.y;
    out[5 * runs + r] = ep.z;
    out[6 * runs + r] = s.vel.x - ref_end[6];
    out[7 * runs + r] = s.vel.y - ref_end[7];
    out[8 * runs + r] = s.vel.z - ref_end[8];
}

__device__ __forceinline__ void store_end(double* __restrict__ out, int64_t runs, int64_t r, const Nav& s) {
    const params_ptr kp = kernarg_params();
    const double ref_end[9] = {kp->ref_end[0], kp->ref_end[1], kp->ref_end[2], kp->ref_end[3], kp->ref_end[4],
                               kp->ref_end[5], kp->ref_end[6], kp->ref_end[7], kp->ref_end[8]};
    // array_error(angle=True) on the last sample: ins_data_manager.py:537-541
    out[0 * runs + r] = angle_range_pi(s.att.yaw - ref_end[0]);
    out[1 * runs + r] = angle_range_pi(s.att.pit - ref_end[1]);
    out[2 * runs + r] = angle_range_pi(s.att.rol - ref_end[2]);
    Vec3 ep{s.pos.x - ref_end[3], s.pos.y - ref_end[4], s.pos.z - ref_end[5]};
    if (kp->end_pos_ned && kp->ref_frame == 0) ep = lla_error_ned(s.pos, Vec3{ref_end[3], ref_end[4], ref_end[5]});
    out[3 * runs + r] = ep.x;
    out[4 * runs + r] = ep.y;
    out[5 * runs + r] = ep.z;
    out[6 * runs + r] = s.vel.x - ref_end[6];
    out[7 * runs + r] = s.vel.y - ref_end[7];
    out[8 * runs + r] = s.vel.z - ref_end[8];
}

// attitude.angle_range_pi with the division by 2 pi replaced by a multiplication, exactly as process_stats_kernel
// (stats.hip) evaluates it: the two statistics paths agree to the bit
__device__ __forceinline__ double angle_range_pi_mul(double x) {
    double m = x - kTwoPi * floor(x * (1.0 / kTwoPi));
    if (m >= kTwoPi) m -= kTwoPi;
    if (m < 0.0) m += kTwoPi;
    return m > kPi ? m - kTwoPi : m;
}

// Online process-error statistics of one run (InsDataMgr.__process_error_stats, ins_data_manager.py:761-795, on
// array_error :519-553): max|e|, mean and std(ddof=0) of the nine error components over the samples >= proc_first, without
// the samples ever leaving the registers -- which is what makes the statistics available when the trajectories are not kept.
//
// Round 3: RAW sums (sum e, sum e^2) instead of round 2's Welford recurrence (a Newton reciprocal and five dependent fp64
// operations per component and step; now one add, one fused multiply-add, one max).  Conditioning: every run starts on the
// truth, so e is the drift accumulated since sample 0 and |mean| is of the order of the std; the variance comes out as
// sum e^2 / n - mean^2 with a relative rounding error of ~ 2^-53 (1 + mean^2 / var) sqrt(n) -- 1e-12 for mean^2 / var up
// to 1e3 at n = 2e5.  process_stats_kernel (stats.hip), which reads kept trajectories, keeps the Welford / Chan-merge form
// and is the checker: the two agree to 1e-9 relative (tests/test_process_stats.py), the online form to 1e-7 with the oracle.
// The floor of the raw form: an error that is (nearly) CONSTANT over the window -- a noise-free or ideal IMU with an initial
// offset, a deterministic bias -- has var << mean^2, and what sum e^2 / n - mean^2 leaves of a std below ~1.5e-8 |mean| is
// rounding (clamped at 0 here).  No variant runs the raw form any more: the sums are kept about an error close to the mean,
// per lane where the registers are there, per launch where they are not (see Proc;
// tests/test_process_stats.py::test_online_statistics_floor_for_a_constant_error).
// The attitude error is wrapped to [-pi, pi] only when some lane of the wavefront is outside it (wrap_pi3).
__device__ __forceinline__ double wrap_pi_lane(double x) { return fabs(x) <= kPi ? x : angle_range_pi_mul(x); }

__device__ __forceinline__ void wrap_pi3(double (&e)[9]) {
    const bool out = !(fabs(e[0]) <= kPi) || !(fabs(e[1]) <= kPi) || !(fabs(e[2]) <= kPi);
    if (__builtin_amdgcn_ballot_w64(out) != 0) {
        e[0] = wrap_pi_lane(e[0]); e[1] = wrap_pi_lane(e[1]); e[2] = wrap_pi_lane(e[2]);
    }
}

// Truth samples are the same for every lane.  Reading them through the constant address space tells the
// compiler the data are invariant, so a wave-uniform index becomes an s_load (scalar cache, lgkmcnt) instead
// of a per-lane global_load -- which matters beyond the 64x fewer bytes: vector loads share the vmcnt counter
// with the trajectory stores, and the s_waitcnt vmcnt(0) guarding them drained every outstanding store once
// per step (measured: +0.45 ms at 65 536 runs).
typedef const double __attribute__((address_space(4))) * uniform_ptr;
__device__ __forceinline__ uniform_ptr as_uniform(const double* p) {
    return (uniform_ptr)(uintptr_t)p;
}

// SHIFT 1 (round 5): the sums are kept about the FIRST in-window error of the run (sum (e - e0), sum (e - e0)^2): a (nearly)
// constant error then leaves var = sum d^2 / n - (sum d / n)^2 with d of the size of the error's VARIATION, and the floor of the
// raw form (~1.5e-8 |mean| on the std) is gone.  Nine more doubles per lane: every process-statistics variant that has them.
// SHIFT 2 (round 6): the ref_frame 0 free-integration variants (245-251 VGPRs: C3's kernel) and the vibration variants do not.
// Their sums are kept about ONE error for the whole launch: that of the first run's initial state against the truth at sample 0
// (proc_shift_kernel writes the nine numbers before the launch; proc_first > 0 keeps sample 0's, the errors grow from it).  It is
// the same for every lane, so it costs no vector register: it is re-read with scalar loads next to the nine subtractions, the way
// the truth sample is.  What is left under the sums is the error's growth plus what the runs' initial states differ by.
// The nine subtractions cost C3 2.3 %; a caller whose runs start ON the truth (nine zero shifts: the same sums either way) may say
// so (ginsim_mc_params.proc_plain_sums, which ginsim.MonteCarloJob works out from the initial state and the truth it uploads) and
// gets SHIFT 0 in these variants.
template <int SHIFT>
struct Proc {
    double s1[9], s2[9], mx[9], e0[SHIFT == 1 ? 9 : 1];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < 9; ++c) { s1[c] = 0.0; s2[c] = 0.0; mx[c] = 0.0; }
#pragma unroll
        for (int c = 0; c < (SHIFT == 1 ? 9 : 1); ++c) e0[c] = 0.0;
    }
    // the process error of one state against the truth sample t = att3, pos3, vel3 (ins_data_manager.py:761-795); NED: the
    // position error in local NED metres (:542-552)
    template <bool NED>
    static __device__ __forceinline__ void error(const Nav& s, const double (&t)[9], double (&e)[9]) {
        e[0] = s.att.yaw - t[0]; e[1] = s.att.pit - t[1]; e[2] = s.att.rol - t[2];
        wrap_pi3(e);
        if (NED) {
            const Vec3 d = lla_error_ned(s.pos, Vec3{t[3], t[4], t[5]});
            e[3] = d.x; e[4] = d.y; e[5] = d.z;
        } else {
            e[3] = s.pos.x - t[3]; e[4] = s.pos.y - t[4]; e[5] = s.pos.z - t[5];
        }
        e[6] = s.vel.x - t[6]; e[7] = s.vel.y - t[7]; e[8] = s.vel.z - t[8];
    }
    // t (wave-uniform): the truth of this sample; first (wave-uniform): this is the first sample of the window;
    // about (SHIFT 2): the launch's nine shifts, wave-uniform
    template <bool NED>
    __device__ __forceinline__ void add(const Nav& s, const double (&t)[9], bool first, uniform_ptr about) {
        double e[9];
        error<NED>(s, t, e);
        if (SHIFT == 1 && first) {
#pragma unroll
            for (int c = 0; c < 9; ++c) e0[c] = e[c];
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const double d = SHIFT == 1 ? e[c] - e0[c] : (SHIFT == 2 ? e[c] - about[c] : e[c]);
            s1[c] += d;
            s2[c] = __builtin_fma(d, d, s2[c]);
            mx[c] = fmax(mx[c], fabs(e[c]));
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ out, int64_t runs, int64_t r, double cnt, uniform_ptr about) const {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const double md = cnt > 0.0 ? s1[c] / cnt : 0.0;
            const double var = cnt > 0.0 ? s2[c] / cnt - md * md : 0.0;
            out[(0 * 9 + c) * runs + r] = mx[c];
            out[(1 * 9 + c) * runs + r] = SHIFT == 1 ? e0[c] + md : (SHIFT == 2 ? about[c] + md : md);
            out[(2 * 9 + c) * runs + r] = var > 0.0 ? sqrt(var) : 0.0;
        }
    }
};

// Sensor sample j of one 3-axis sensor: truth + bias + drift + white  (pathgen.py:500, 562), and the
// Gauss-Markov update d[j+1] = a d[j] + b N[j] (pathgen.py:589-590).
__device__ __forceinline__ Vec3 load3(uniform_ptr ref, int64_t j) { return Vec3{ref[3 * j], ref[3 * j + 1], ref[3 * j + 2]}; }

// WD = false: the launcher saw no axis with an infinite correlation time (white_drift) and no constant bias in either
// sensor -- every standard IMU grade of imu_model.py -- so the six wave-uniform selects and the three bias additions
// per sensor are compiled out (x + 0.0 == x: the values are the same).
template <bool WD = true>
__device__ __forceinline__ Vec3 sense3(const Vec3& truth, model_ptr m, Vec3& drift, const Vec3& zd,
                                       const Vec3& zw) {
    const double bx = m->gm_b[0] * zd.x, by = m->gm_b[1] * zd.y, bz = m->gm_b[2] * zd.z;
    const double dx = (WD && m->white_drift[0]) ? bx : drift.x;
    const double dy = (WD && m->white_drift[1]) ? by : drift.y;
    const double dz = (WD && m->white_drift[2]) ? bz : drift.z;
    Vec3 o;
    if (WD) {
        o.x = truth.x + m->bias[0] + dx + m->white[0] * zw.x;
        o.y = truth.y + m->bias[1] + dy + m->white[1] * zw.y;
        o.z = truth.z + m->bias[2] + dz + m->white[2] * zw.z;
    } else {
        o.x = truth.x + dx + m->white[0] * zw.x;
        o.y = truth.y + dy + m->white[1] * zw.y;
        o.z = truth.z + dz + m->white[2] * zw.z;
    }
    drift.x = __builtin_fma(m->gm_a[0], drift.x, bx);
    drift.y = __builtin_fma(m->gm_a[1], drift.y, by);
    drift.z = __builtin_fma(m->gm_a[2], drift.z, bz);
    return o;
}

// The vibration term of a sensor sample (ABI 5; pathgen.py:476-492, 538-556), added LAST as the reference's sum does
// (a_mea = ref + bias + drift + noise + vib).  Everything wave-uniform except the normals and the per-run phases.
typedef const ginsim_vibration __attribute__((address_space(4))) * vib_ptr;

template <uint32_t PHASE_STREAM>
__device__ __forceinline__ Vec3 vibration_phase(vib_ptr v, const RngKey& key) {
    if (v->type != GINSIM_VIB_SINUSOIDAL || !v->random_phase) return Vec3{0.0, 0.0, 0.0};
    const u32x4 w = philox4x32(0u, PHASE_STREAM >> 1, key.r0, key.r1, key.k0, key.k1);
    // np.random.rand(1)*2*math.pi (pathgen.py:553-555), u = word 2^-32
    return Vec3{((double)w.x * 0x1p-32 * 2.0) * kPi, ((double)w.y * 0x1p-32 * 2.0) * kPi, ((double)w.z * 0x1p-32 * 2.0) * kPi};
}

// the term itself, the three normals of a 'random' vibration given (zx, zy, zz: whoever generated them -- this wavefront, or the
// producers of the wave-specialised kernel through the LDS ring)
__device__ __forceinline__ Vec3 vibration_term(const Vec3& o, vib_ptr v, double zx, double zy, double zz, uint32_t j, const Vec3& phase) {
    Vec3 r = o;
    if (v->type == GINSIM_VIB_RANDOM) {
        r.x = o.x + v->amp[0] * zx;
        r.y = o.y + v->amp[1] * zy;
        r.z = o.z + v->amp[2] * zz;
    } else if (v->type == GINSIM_VIB_SINUSOIDAL) {
        const double cj = v->omega_dt * (double)j;          // (2 pi f dt) * arange(n), rounded before the phase is added
        if (v->random_phase) {
            const double ax = cj + phase.x, ay = cj + phase.y, az = cj + phase.z;
            r.x = o.x + v->amp[0] * sin(ax);
            r.y = o.y + v->amp[1] * sin(ay);
            r.z = o.z + v->amp[2] * sin(az);
        } else {
            const double s = sin(cj);
            r.x = o.x + v->amp[0] * s;
            r.y = o.y + v->amp[1] * s;
            r.z = o.z + v->amp[2] * s;
        }
    }
    return r;
}

// The three values of sample j of a 'psd' vibration (ABI 8): the series were made before the launch (vib_psd.hip), [axis][sample mod
// period][run], tiled to n (time_series_from_psd.py:58-63).  Zero for every other type.  Called at the TOP of a step, ~800
// instructions before the sum that takes the values: asked for next to the sum, the loads cost their whole latency every step.
__device__ __forceinline__ Vec3 psd_vibration(vib_ptr v, uint32_t j, int64_t run, int64_t runs) {
    if (v->type != GINSIM_VIB_PSD) return Vec3{0.0, 0.0, 0.0};
    const int64_t period = v->period;
    const double* s = v->series + (int64_t)(j % (uint32_t)period) * runs + run;
    const int64_t pl = period * runs;
    return Vec3{__builtin_nontemporal_load(s), __builtin_nontemporal_load(s + pl), __builtin_nontemporal_load(s + 2 * pl)};
}

// psd: psd_vibration() of this sensor and sample
template <uint32_t STREAM>
__device__ __forceinline__ Vec3 add_vibration(const Vec3& o, vib_ptr v, const RngKey& key, uint32_t j, const NormalTables& tab,
                                              const Vec3& phase, const Vec3& psd) {
    Vec3 r = o;
    if (v->type == GINSIM_VIB_PSD) {
        r.x = o.x + psd.x;
        r.y = o.y + psd.y;
        r.z = o.z + psd.z;
    } else if (v->type == GINSIM_VIB_RANDOM) {
        double z0[2], z1[2];
        normal_pairs<STREAM, 2>(key, j, z0, z1, tab);
        r.x = o.x + v->amp[0] * z0[0];
        r.y = o.y + v->amp[1] * z1[0];
        r.z = o.z + v->amp[2] * z0[1];
    } else if (v->type == GINSIM_VIB_SINUSOIDAL) {
        const double cj = v->omega_dt * (double)j;          // (2 pi f dt) * arange(n), rounded before the phase is added
        if (v->random_phase) {
            const double ax = cj + phase.x, ay = cj + phase.y, az = cj + phase.z;
            r.x = o.x + v->amp[0] * sin(ax);
            r.y = o.y + v->amp[1] * sin(ay);
            r.z = o.z + v->amp[2] * sin(az);
        } else {
            const double s = sin(cj);
            r.x = o.x + v->amp[0] * s;
            r.y = o.y + v->amp[1] * s;
            r.z = o.z + v->amp[2] * s;
        }
    }
    return r;
}

// Two workgroups per CU is what the launch geometry below counts on: tell the register allocator (variants had grown
// to 256 VGPRs + a few AGPRs = one wavefront per SIMD, 30 % slower at 262 144 runs, with no functional symptom;
// tests/test_host_cpu.py now reads the compiler's resource report).
// PS: 0 = no process statistics; 1 = online process-error statistics of the (single) algorithm; 2 = the same with the
// position error in NED metres (ref_frame 0); 3 / 4 = 1 / 2 with the sums taken as they are (ginsim_mc_params.proc_plain_sums:
// the caller states that the runs start ON the truth, so the launch's shift would be nine zeros -- the same sums without the
// nine subtractions per step; only the variants whose shift is per launch have the form, see Proc).
// VIB: the sensors carry a vibration term (vib_accel / vib_gyro; general sensor model, generate mode).
template <int RF, int ALGOS, bool GIVEN, bool WD, int PS = 0, bool VIB = false>
__global__ void __launch_bounds__(256, 2) mc_kernel(const ginsim_mc_params a, const double* __restrict__ proc_about) {
    static_assert(PS == 0 || (!GIVEN && (ALGOS == GINSIM_ALGO_FREE || ALGOS == GINSIM_ALGO_ODO)), "process statistics: one algorithm, generate mode");
    static_assert(!VIB || (!GIVEN && WD), "vibration: generate mode, general sensor model");
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t* trace = nullptr;
    if (a.wave_trace && (threadIdx.x & 63) == 0) {
        trace = a.wave_trace + 4 * (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        trace[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        trace[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        trace[2] = __builtin_amdgcn_s_memtime();
    }
    __shared__ uint32_t ntab[GIVEN ? 4 : kNormalLdsWords];
    NormalTables tab{};
    if (!GIVEN) {
        tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
        __syncthreads();
    }
    if (r >= a.runs) return;
    constexpr bool FREE = (ALGOS & GINSIM_ALGO_FREE) != 0;
    constexpr bool ODO = (ALGOS & GINSIM_ALGO_ODO) != 0;
    const int64_t n = a.n;
    const int64_t runs = a.runs;
    const int64_t plane = n * runs;
    const double dt = 1.0 / a.fs;

    // which set of initial states: free_integration.py:85-87 (run_times counts calls since construction)
    const uint64_t call = a.ini_first + (uint64_t)r;
    const double* ini = a.ini + 10 * (call < (uint64_t)a.n_ini ? call : 0);
    Nav fi, od;
    if (FREE) nav_init<RF>(fi, ini, a.ini_has_g);
    if (ODO) nav_init<RF>(od, ini, a.ini_has_g);

    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    Vec3 da{0.0, 0.0, 0.0}, dg{0.0, 0.0, 0.0};
    Vec3 vpa{0.0, 0.0, 0.0}, vpg{0.0, 0.0, 0.0};
    if (VIB) {
        vpa = vibration_phase<S_ACC_VIB_PHASE>(&kernarg_params()->vib_accel, key);
        vpg = vibration_phase<S_GYR_VIB_PHASE>(&kernarg_params()->vib_gyro, key);
    }
    MathConsts mk;
    // constants pinned in VGPRs except where that variant would spill to scratch (measured per variant)
    mk.init<(ALGOS != (GINSIM_ALGO_FREE | GINSIM_ALGO_ODO)) && PS == 0 && !VIB>();

    if (FREE && a.out_traj[0]) store9(a.out_traj[0], plane, r, fi);
    if (ODO && a.out_traj[1]) store9(a.out_traj[1], plane, r, od);
    // the sums shifted per lane wherever the registers are there, per launch elsewhere (see Proc)
    constexpr bool PNED = PS == 2 || PS == 4, PPLAIN = PS >= 3;
    static_assert(!PPLAIN || (RF == 0 && ALGOS == GINSIM_ALGO_FREE && !VIB), "plain sums: only where the shift is per launch and optional");
    constexpr int PSHIFT = PS == 0 ? 0 : ((!VIB && !(RF == 0 && ALGOS == GINSIM_ALGO_FREE)) ? 1 : (PPLAIN ? 0 : 2));
    const uniform_ptr nav_truth = as_uniform(a.ref_nav);
    // opaque per use: hoisted out of the time loop the nine shifts would sit in 18 SGPRs the loop does not have
    auto about = [&]() -> uniform_ptr {
        uniform_ptr q = as_uniform(proc_about);
        if (PSHIFT == 2) asm volatile("" : "+s"(q));
        return q;
    };
    Proc<PSHIFT> ps;
    if (PS) {
        ps.clear();
        if (a.proc_first <= 0) {                    // sample 0 is the initial state (free_integration.py:96-102)
            const double t[9] = {nav_truth[0], nav_truth[1], nav_truth[2], nav_truth[3], nav_truth[4], nav_truth[5],
                                 nav_truth[6], nav_truth[7], nav_truth[8]};
            ps.template add<PNED>(FREE ? fi : od, t, true, about());
        }
    }

    for (int64_t j = 0; j < n; ++j) {
        const int64_t off = j * runs + r;
        Vec3 acc, gyr;
        double odo = 0.0;
        if (GIVEN) {
            if (j == n - 1) break;
            gyr = Vec3{a.in_gyro[off], a.in_gyro[plane + off], a.in_gyro[2 * plane + off]};
            if (FREE) acc = Vec3{a.in_accel[off], a.in_accel[plane + off], a.in_accel[2 * plane + off]};
            if (ODO) odo = a.in_odo[off];
        } else {
            const bool last = (j == n - 1);
            // the last sample only exists as sensor output; skip it when nothing stores it
            if (last && !a.out_accel && !a.out_gyro && !a.out_odo) break;
            const uint32_t jj = (uint32_t)j;
            // wave-uniform truth of this step: requested here, ~800 instructions before the sensor sums use it
            const Vec3 cur_a = load3(as_uniform(a.ref_accel), j), cur_g = load3(as_uniform(a.ref_gyro), j);
            Vec3 psd_a{0.0, 0.0, 0.0}, psd_g{0.0, 0.0, 0.0};
            if (VIB) {
                psd_a = psd_vibration(&kernarg_params()->vib_accel, jj, r, runs);
                psd_g = psd_vibration(&kernarg_params()->vib_gyro, jj, r, runs);
            }
            const bool need_acc = FREE || a.out_accel;
            const bool need_gyr = FREE || ODO || a.out_gyro;
            const bool need_odo = ODO || a.out_odo;
            if (need_acc && need_gyr) {             // the common case: six streams in one phased batch
                double z0[6], z1[6];
                normal_pairs<S_ACC_D_XY, 6>(key, jj, z0, z1, tab);
                const params_ptr kp = kernarg_params();
                acc = sense3<WD>(cur_a, &kp->accel, da, Vec3{z0[0], z1[0], z0[1]}, Vec3{z1[1], z0[2], z1[2]});
                gyr = sense3<WD>(cur_g, &kp->gyro, dg, Vec3{z0[3], z1[3], z0[4]}, Vec3{z1[4], z0[5], z1[5]});
            } else if (need_acc) {
                double z0[3], z1[3];
                normal_pairs<S_ACC_D_XY, 3>(key, jj, z0, z1, tab);
                acc = sense3<WD>(cur_a, &kernarg_params()->accel, da, Vec3{z0[0], z1[0], z0[1]}, Vec3{z1[1], z0[2], z1[2]});
            } else if (need_gyr) {
                double z0[3], z1[3];
                normal_pairs<S_GYR_D_XY, 3>(key, jj, z0, z1, tab);
                gyr = sense3<WD>(cur_g, &kernarg_params()->gyro, dg, Vec3{z0[0], z1[0], z0[1]}, Vec3{z1[1], z0[2], z1[2]});
            }
            if (VIB) {
                if (need_acc) acc = add_vibration<S_ACC_VIB_XY>(acc, &kernarg_params()->vib_accel, key, jj, tab, vpa, psd_a);
                if (need_gyr) gyr = add_vibration<S_GYR_VIB_XY>(gyr, &kernarg_params()->vib_gyro, key, jj, tab, vpg, psd_g);
            }
            if (need_acc && a.out_accel) store3(a.out_accel, plane, off, acc);
            if (need_gyr && a.out_gyro) store3(a.out_gyro, plane, off, gyr);
            if (need_odo) {
                double z0, z1;
                normal_pair(key, S_ODO, jj, z0, z1, tab);
                const params_ptr kq = kernarg_params();
                odo = kq->odo_scale * as_uniform(a.ref_odo)[j] + kq->odo_stdv * z0;     // pathgen.py:639-640
                if (a.out_odo) a.out_odo[off] = odo;
            }
            if (last) break;
        }
        const bool resync = ((j + 1) & (kTrigResync - 1)) == 0;
        if (FREE) {
            nav_step<RF, false, !GIVEN>(fi, gyr, acc, 0.0, dt, a.earth_rot, resync, mk);
            if (a.out_traj[0]) store9(a.out_traj[0], plane, off + runs, fi);
        }
        if (ODO) {
            nav_step<RF, true, !GIVEN>(od, gyr, acc, odo, dt, a.earth_rot, resync, mk);
            if (a.out_traj[1]) store9(a.out_traj[1], plane, off + runs, od);
        }
        if (PS) {
            if (j + 1 >= a.proc_first) {            // wave-uniform
                const uniform_ptr q = nav_truth + 9 * (j + 1);
                const double t[9] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]};
                ps.template add<PNED>(FREE ? fi : od, t, a.proc_first > 0 && j + 1 == a.proc_first, about());
            }
        }
    }
    if (FREE && a.out_end[0]) store_end(a.out_end[0], runs, r, fi);
    if (ODO && a.out_end[1]) store_end(a.out_end[1], runs, r, od);
    if (RF == 0) {
        if (FREE && a.out_end_ned[0]) store_end_ned(a.out_end_ned[0], runs, r, fi);
        if (ODO && a.out_end_ned[1]) store_end_ned(a.out_end_ned[1], runs, r, od);
    }
    if (PS) ps.store(a.out_proc[FREE ? 0 : 1], runs, r, (double)(n - (a.proc_first > 0 ? a.proc_first : 0)), about());
    if (trace) trace[3] = __builtin_amdgcn_s_memtime();
}

// The launch-wide shift of the process statistics (Proc, SHIFT 2): the process error of the FIRST run's initial state against
// the truth at sample 0, nine doubles.  One lane, before the launch, on its stream.
template <int RF, bool NED>
__global__ void proc_shift_kernel(const ginsim_mc_params a, double* __restrict__ about) {
    const double* ini = a.ini + 10 * (a.ini_first < (uint64_t)a.n_ini ? a.ini_first : 0);
    Nav z;
    nav_init<RF>(z, ini, a.ini_has_g);
    const double t[9] = {a.ref_nav[0], a.ref_nav[1], a.ref_nav[2], a.ref_nav[3], a.ref_nav[4], a.ref_nav[5],
                         a.ref_nav[6], a.ref_nav[7], a.ref_nav[8]};
    double e[9];
    Proc<0>::error<NED>(z, t, e);
    for (int c = 0; c < 9; ++c) about[c] = e[c];
}

// ---------------------------------------------------------------------------------------------------
// Wave-specialised variant, written for SMALL batches (<= 1024 wavefronts of runs, i.e. one wavefront per SIMD with the
// kernel above -- BASELINE config 2) and, with two producer groups, the faster one at every size (mc_variant).  A lone wavefront cannot hide its own dependent-instruction and s_waitcnt latencies
// and there are no more runs to give the SIMD a second wavefront.  So the work of one step is split across TWO
// wavefronts per 64 runs, at the point where the normal generator changes character:
//
//   waves 4-7 of a 512-thread workgroup (producers): the three Philox blocks of a step and the twelve single-precision
//                                                    normal transforms -> LDS ring, per step and run the twelve
//                                                    normals as floats (48 B)
//   waves 0-3 (consumers)                          : read tile i-1 from LDS, widen, sensor sums, mechanisation, all stores
//
// One __syncthreads() per tile of T = kSplitTile (6) steps; the instruction total is unchanged, the SIMD just always has a second
// wavefront to issue from.  Results are bit-identical to mc_kernel (same functions in the same order on the same values).
constexpr int kSplitRuns = 256;
#ifndef GINSIM_SPLIT_TILE
#define GINSIM_SPLIT_TILE 6
#endif
constexpr int kSplitTile = GINSIM_SPLIT_TILE;
constexpr int kSplitStep = 12 * 4;                      // bytes per step and run in the ring
// 96 KiB of ring + the tables, padded to more than half of the LDS so that a CU takes ONE workgroup (eight wavefronts, two
// per SIMD: a producer and a consumer)
constexpr size_t kSplitRing = (size_t)2 * kSplitTile * kSplitStep * kSplitRuns;
constexpr size_t kSplitLds = kSplitRing > 100 * 1024 ? kSplitRing : 100 * 1024;

// PROD producer wavefronts per consumer wavefront: with two (768 threads, three wavefronts per SIMD, <= 168 registers) the
// steps of a tile alternate between the two producer groups.
// KEEP = false: a statistics-only launch (no series pointer set): the store code and its address registers are compiled out,
// which is what lets the ref_frame 0 consumer fit the 168 registers of three wavefronts per SIMD.
// VIB (round 5): the vibration term of Sim(env=...) in the wave-specialised kernel, for the batches where the plain vibration kernel
// runs with ONE wavefront per SIMD (<= 1024 wavefronts of runs: C2).  The three normals per sensor of a 'random' vibration are one
// more Philox block each and come from the producers -- the ring carries 18 floats per step and run instead of 12, in tiles of 4
// steps instead of 6 (the same 144 KB) --, a sinusoidal term is evaluated by the consumer; same operations on the same values as
// mc_kernel<..., VIB = true> (vibration_term), so the two kernels agree to the bit.
template <int RF, int ALGOS, bool WD, int PROD = 1, bool KEEP = true, bool VIB = false>
__global__ void __launch_bounds__(256 * (1 + PROD)) mc_kernel_split(const ginsim_mc_params a_in) {
    static_assert(!VIB || WD, "vibration: general sensor model");
    constexpr int kSplitTile = VIB ? 4 : ginsim::kSplitTile;
    ginsim_mc_params a = a_in;
    if (!KEEP) {
        a.out_accel = a.out_gyro = a.out_odo = nullptr;
        a.out_traj[0] = a.out_traj[1] = nullptr;
    }
    extern __shared__ float zring[];                    // [2 stages][T steps][12 normals][256 runs]
    constexpr bool FREE = (ALGOS & GINSIM_ALGO_FREE) != 0;
    constexpr bool ODO = (ALGOS & GINSIM_ALGO_ODO) != 0;
    constexpr int kStepFloats = (VIB ? 18 : 12) * kSplitRuns;          // 3072 (4608) floats per step
    const int lane = threadIdx.x & (kSplitRuns - 1);
    const bool producer = threadIdx.x >= kSplitRuns;
    const int pgroup = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) - 1;     // which producer group (wave-uniform)
    const int64_t r = (int64_t)blockIdx.x * kSplitRuns + lane;
    const bool active = r < a.runs;
    const int64_t n = a.n, runs = a.runs, plane = n * runs;
    const bool keep_last = a.out_accel || a.out_gyro || a.out_odo;      // the last sample only exists as sensor output
    const int64_t n_noise = keep_last ? n : n - 1;
    const int64_t ntiles = (n_noise + kSplitTile - 1) / kSplitTile;
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    MathConsts mk;
    mk.init<(ALGOS != (GINSIM_ALGO_FREE | GINSIM_ALGO_ODO))>();     // the two-algorithm consumer would spill
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();


    if (producer) {
        for (int64_t i = 0; i <= ntiles; ++i) {
            if (i < ntiles && active) {
                float* stage = zring + (i & 1) * (kSplitTile * kStepFloats);
#pragma unroll
                for (int t = 0; t < kSplitTile; ++t) {
                    const int64_t j = i * kSplitTile + t;
                    if (j < n_noise && (PROD == 1 || (t % PROD) == pgroup)) {
                        float z0[6], z1[6];
                        normal_pairs_f32<S_ACC_D_XY, 6>(key, (uint32_t)j, z0, z1, tab);
                        float* zb = stage + t * kStepFloats + lane;
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            zb[(2 * k) * kSplitRuns] = z0[k];
                            zb[(2 * k + 1) * kSplitRuns] = z1[k];
                        }
                        if (VIB) {              // streams 10-13: one block per sensor with a 'random' vibration (wave-uniform)
                            const params_ptr kv = kernarg_params();
                            if (kv->vib_accel.type == GINSIM_VIB_RANDOM) {
                                float v0[2], v1[2];
                                normal_pairs_f32<S_ACC_VIB_XY, 2>(key, (uint32_t)j, v0, v1, tab);
                                zb[12 * kSplitRuns] = v0[0]; zb[13 * kSplitRuns] = v1[0]; zb[14 * kSplitRuns] = v0[1];
                            }
                            if (kv->vib_gyro.type == GINSIM_VIB_RANDOM) {
                                float v0[2], v1[2];
                                normal_pairs_f32<S_GYR_VIB_XY, 2>(key, (uint32_t)j, v0, v1, tab);
                                zb[15 * kSplitRuns] = v0[0]; zb[16 * kSplitRuns] = v1[0]; zb[17 * kSplitRuns] = v0[1];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    const double dt = 1.0 / a.fs;
    const uint64_t call = a.ini_first + (uint64_t)r;
    const double* ini = a.ini + 10 * ((active && call < (uint64_t)a.n_ini) ? call : 0);
    Nav fi, od;
    if (FREE) nav_init<RF>(fi, ini, a.ini_has_g);
    if (ODO) nav_init<RF>(od, ini, a.ini_has_g);
    Vec3 da{0.0, 0.0, 0.0}, dg{0.0, 0.0, 0.0};
    Vec3 vpa{0.0, 0.0, 0.0}, vpg{0.0, 0.0, 0.0};
    if (VIB) {
        vpa = vibration_phase<S_ACC_VIB_PHASE>(&kernarg_params()->vib_accel, key);
        vpg = vibration_phase<S_GYR_VIB_PHASE>(&kernarg_params()->vib_gyro, key);
    }
    if (active) {
        if (FREE && a.out_traj[0]) store9(a.out_traj[0], plane, r, fi);
        if (ODO && a.out_traj[1]) store9(a.out_traj[1], plane, r, od);
    }
    for (int64_t i = 0; i <= ntiles; ++i) {
        if (i >= 1 && active) {
            const float* stage = zring + ((i - 1) & 1) * (kSplitTile * kStepFloats);
#pragma unroll 1
            for (int t = 0; t < kSplitTile; ++t) {
                const int64_t j = (i - 1) * kSplitTile + t;
                if (j >= n_noise) break;
                const int64_t off = j * runs + r;
                const bool last = (j == n - 1);
                const Vec3 cur_a = load3(as_uniform(a.ref_accel), j), cur_g = load3(as_uniform(a.ref_gyro), j);
                const float* zb = stage + t * kStepFloats + lane;
                double p0[6], p1[6];                  // z0 / z1 of streams 0..5
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    p0[k] = (double)zb[(2 * k) * kSplitRuns];
                    p1[k] = (double)zb[(2 * k + 1) * kSplitRuns];
                }
                const params_ptr kp = kernarg_params();
                Vec3 acc = sense3<WD>(cur_a, &kp->accel, da, Vec3{p0[0], p1[0], p0[1]}, Vec3{p1[1], p0[2], p1[2]});
                Vec3 gyr = sense3<WD>(cur_g, &kp->gyro, dg, Vec3{p0[3], p1[3], p0[4]}, Vec3{p1[4], p0[5], p1[5]});
                if (VIB) {              // added last, as pathgen.py:500, 562 do
                    const params_ptr kv = kernarg_params();
                    double va[3] = {0.0, 0.0, 0.0}, vg[3] = {0.0, 0.0, 0.0};
                    if (kv->vib_accel.type == GINSIM_VIB_RANDOM) {
                        va[0] = (double)zb[12 * kSplitRuns]; va[1] = (double)zb[13 * kSplitRuns]; va[2] = (double)zb[14 * kSplitRuns];
                    }
                    if (kv->vib_gyro.type == GINSIM_VIB_RANDOM) {
                        vg[0] = (double)zb[15 * kSplitRuns]; vg[1] = (double)zb[16 * kSplitRuns]; vg[2] = (double)zb[17 * kSplitRuns];
                    }
                    acc = vibration_term(acc, &kv->vib_accel, va[0], va[1], va[2], (uint32_t)j, vpa);
                    gyr = vibration_term(gyr, &kv->vib_gyro, vg[0], vg[1], vg[2], (uint32_t)j, vpg);
                }
                if (a.out_accel) store3(a.out_accel, plane, off, acc);
                if (a.out_gyro) store3(a.out_gyro, plane, off, gyr);
                double odo = 0.0;
                if (ODO || a.out_odo) {
                    double z0, z1;
                    normal_pair(key, S_ODO, (uint32_t)j, z0, z1, tab);
                    const params_ptr kq = kernarg_params();
                    odo = kq->odo_scale * as_uniform(a.ref_odo)[j] + kq->odo_stdv * z0;
                    if (a.out_odo) a.out_odo[off] = odo;
                }
                if (last) break;
                const bool resync = ((j + 1) & (kTrigResync - 1)) == 0;
                if (FREE) {
                    nav_step<RF, false>(fi, gyr, acc, 0.0, dt, a.earth_rot, resync, mk);
                    if (a.out_traj[0]) store9(a.out_traj[0], plane, off + runs, fi);
                }
                if (ODO) {
                    nav_step<RF, true>(od, gyr, acc, odo, dt, a.earth_rot, resync, mk);
                    if (a.out_traj[1]) store9(a.out_traj[1], plane, off + runs, od);
                }
            }
        }
        __syncthreads();
    }
    if (active) {
        if (FREE && a.out_end[0]) store_end(a.out_end[0], runs, r, fi);
        if (ODO && a.out_end[1]) store_end(a.out_end[1], runs, r, od);
        if (RF == 0) {
            if (FREE && a.out_end_ned[0]) store_end_ned(a.out_end_ned[0], runs, r, fi);
            if (ODO && a.out_end_ned[1]) store_end_ned(a.out_end_ned[1], runs, r, od);
        }
    }
}

// Launch geometry.  The kernel is VALU-bound and every wavefront of a launch does the same amount of work,
// so the only thing that matters is that wavefronts are spread evenly over the 1024 SIMDs.  Measured on
// MI355X: with 64-thread workgroups the dispatcher, depending on what ran before, doubles up ~6 % of the
// SIMDs and leaves as many idle (7.2 ms instead of 4.2 ms at 65 536 runs).  256-thread workgroups put one
// wavefront on each SIMD of a CU, and a dynamic-LDS reservation (never touched by the kernel) caps the
// workgroups per CU at k = 1 (<= 1024 wavefronts) or 2 (the VGPR budget allows no more), so every SIMD
// holds exactly k wavefronts until the tail.
constexpr int kBlock = 256;
constexpr size_t kLdsPerCu = 160 * 1024;

static int split_policy() {        // GINSIM_SPLIT=0 / 1 forces the plain / wave-specialised kernel (A/B measurements)
    static const int v = [] { const char* e = getenv("GINSIM_SPLIT"); return e ? atoi(e) : -1; }();
    return v;
}

// 1 = wave-specialised kernel (mc_kernel_split), 0 = one wavefront does everything for its 64 runs (mc_kernel)
static bool any_vibration(const ginsim_mc_params& p) { return p.vib_accel.type != GINSIM_VIB_NONE || p.vib_gyro.type != GINSIM_VIB_NONE; }
// a 'psd' vibration (ABI 8) is a series read per lane and sample: the lane-per-run kernel has it, nothing else
static bool any_psd_vibration(const ginsim_mc_params& p) { return p.vib_accel.type == GINSIM_VIB_PSD || p.vib_gyro.type == GINSIM_VIB_PSD; }

int mc_variant(const ginsim_mc_params& p) {
    if (any_psd_vibration(p)) return 0;
    if (any_vibration(p)) {
        // the vibration term lives in the plain kernels, except where they would run with one wavefront per SIMD: a single free
        // integration, generated sensors, at most 1024 wavefronts of runs (C2's shape) -> mc_kernel_split<..., VIB = true>
        const char* env = getenv("GINSIM_SPLIT_VIB");        // read per call: the tests run both kernels in one process
        return (env ? atoi(env) : 1) != 0 && split_policy() != 0 && p.algo_mask == GINSIM_ALGO_FREE && !p.given_sensors && p.block_threads == 0 &&
               !p.wave_trace && p.n >= 2 && !(p.out_proc[0] || p.out_proc[1]) && (p.runs + kWave - 1) / kWave <= 1024;
    }
    if (!(p.algo_mask & GINSIM_ALGO_FREE) || p.given_sensors || p.block_threads != 0 || p.wave_trace || p.n < 2) return 0;
    if (p.out_proc[0] || p.out_proc[1]) return 0;      // online process statistics live in the plain kernel
    const int pol = split_policy();
    if (pol >= 0) return pol != 0;
    // one algorithm: three wavefronts per SIMD (a consumer and two producers) beat the plain kernel's two at every size
    // (131 072 runs: 2.71 against 3.06 ms, 262 144: 5.76 against 6.13); two algorithms: only while the plain kernel
    // cannot fill the SIMDs with a second wavefront
    if (p.algo_mask == GINSIM_ALGO_FREE && p.ref_frame == 1) return 1;      // the variants with two producer groups (launch3)
    return (p.runs + kWave - 1) / kWave <= 1024 ? 1 : 0;
}

// white-drift axes or a constant bias anywhere: the general sensor model (WD = true kernels)
static bool any_white_drift(const ginsim_mc_params& p) {
    bool f = false;
    for (int k = 0; k < 3; ++k)
        f = f || p.accel.white_drift[k] || p.gyro.white_drift[k] || p.accel.bias[k] != 0.0 || p.gyro.bias[k] != 0.0;
    return f;
}

// name != nullptr: write the kernel's name (as rocprofv3 reports it, without arguments) instead of launching -- what
// ginsim_mc_kernel_name returns, so that profiles and the bench attribute to the instantiation that really runs
#define GINSIM_NAME_OR(fmt, ...)                         \
    if (name) {                                          \
        snprintf(name, cap, fmt, __VA_ARGS__);           \
        return hipSuccess;                               \
    }
static const char* tf(bool b) { return b ? "true" : "false"; }

template <int RF, int ALGOS, bool WD>
static hipError_t launch3(const ginsim_mc_params& p, hipStream_t stream, char* name, size_t cap, double* about) {
    const int tb = p.block_threads > 0 ? p.block_threads : kBlock;
    const int64_t waves = (p.runs + kWave - 1) / kWave;
    if constexpr ((ALGOS & GINSIM_ALGO_FREE) != 0) {
        if (mc_variant(p) == 1) {
            // one algorithm in the ECEF-free frame fits 168 registers: two producer wavefronts per consumer, three wavefronts
            // per SIMD (C2: 1.48 -> 1.41 ms); ref_frame 0 would spill 76-140 B per lane
            constexpr int PROD = (ALGOS == GINSIM_ALGO_FREE && RF == 1) ? 2 : 1;
            static const int prod = [] { const char* e = getenv("GINSIM_SPLIT_PROD"); return e ? atoi(e) : PROD; }();
            const dim3 sgrid((unsigned)((p.runs + kSplitRuns - 1) / kSplitRuns));
            if constexpr (ALGOS == GINSIM_ALGO_FREE && WD) {
                if (any_vibration(p)) {         // one producer group: the consumer with the vibration term wants more than 168 registers
                    constexpr size_t lds = (size_t)2 * 4 * 18 * 4 * kSplitRuns;
                    GINSIM_NAME_OR("ginsim::mc_kernel_split<%d, %d, %s, 1, true, true>", RF, ALGOS, tf(WD))
                    static PerDeviceOnce oncev;
                    oncev.run([] {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mc_kernel_split<RF, ALGOS, WD, 1, true, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    });
                    hipLaunchKernelGGL((mc_kernel_split<RF, ALGOS, WD, 1, true, true>), sgrid, dim3(512), lds, stream, p);
                    return hipGetLastError();
                }
            }
            if constexpr (ALGOS == GINSIM_ALGO_FREE && RF == 0) {     // nothing kept: two producer groups fit here too
                const bool keep = p.out_accel || p.out_gyro || p.out_odo || p.out_traj[0] || p.out_traj[1];
                if (!keep && prod != 1) {
                    GINSIM_NAME_OR("ginsim::mc_kernel_split<%d, %d, %s, 2, false, false>", RF, ALGOS, tf(WD))
                    static PerDeviceOnce once2;
                    once2.run([] {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mc_kernel_split<RF, ALGOS, WD, 2, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitLds);
                    });
                    hipLaunchKernelGGL((mc_kernel_split<RF, ALGOS, WD, 2, false>), sgrid, dim3(768), kSplitLds, stream, p);
                    return hipGetLastError();
                }
            }
            const bool two = prod == PROD && PROD > 1;
            GINSIM_NAME_OR("ginsim::mc_kernel_split<%d, %d, %s, %d, true, false>", RF, ALGOS, tf(WD), two ? PROD : 1)
            static PerDeviceOnce once;
            once.run([] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mc_kernel_split<RF, ALGOS, WD, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitLds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mc_kernel_split<RF, ALGOS, WD, PROD>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitLds);
            });
            if (two)
                hipLaunchKernelGGL((mc_kernel_split<RF, ALGOS, WD, PROD>), sgrid, dim3(256 * (1 + PROD)), kSplitLds, stream, p);
            else
                hipLaunchKernelGGL((mc_kernel_split<RF, ALGOS, WD, 1>), sgrid, dim3(512), kSplitLds, stream, p);
            return hipGetLastError();
        }
    }
    const int per_cu = waves <= 1024 ? 1 : 2;
    // strictly more than 1/(k+1) of the LDS so that k+1 workgroups do not fit: 81 KB (k = 1), 54 KB (k = 2)
    const size_t lds = p.block_threads > 0 ? 0 : kLdsPerCu / (per_cu + 1) + 1024;
    const dim3 grid((unsigned)((p.runs + tb - 1) / tb)), block(tb);
    if constexpr (ALGOS == GINSIM_ALGO_FREE || ALGOS == GINSIM_ALGO_ODO) {
        if (p.out_proc[ALGOS == GINSIM_ALGO_FREE ? 0 : 1]) {
            const bool ned = RF == 0 && p.proc_pos_ned;
            // the variants whose sums are shifted per launch (Proc, SHIFT 2): the nine shifts are written first, on the same stream
            auto write_shift = [&](bool vib) -> hipError_t {
                if (!vib && !(RF == 0 && ALGOS == GINSIM_ALGO_FREE)) return hipSuccess;
                if (!about) return hipErrorInvalidValue;
                if (ned) hipLaunchKernelGGL((proc_shift_kernel<RF, RF == 0>), dim3(1), dim3(1), 0, stream, p, about);
                else hipLaunchKernelGGL((proc_shift_kernel<RF, false>), dim3(1), dim3(1), 0, stream, p, about);
                return hipGetLastError();
            };
            // ... unless the caller states that the runs start on the truth (nine zero shifts): the plain sums
            constexpr bool kPlainForm = RF == 0 && ALGOS == GINSIM_ALGO_FREE;
            const bool plain = kPlainForm && p.proc_plain_sums != 0 && !any_vibration(p);
            if constexpr (WD) {                     // the general sensor model: every statistics form, vibration included
                if (any_vibration(p)) {
                    GINSIM_NAME_OR("ginsim::mc_kernel<%d, %d, false, true, %d, true>", RF, ALGOS, ned ? 2 : 1)
                    if (hipError_t e = write_shift(true); e != hipSuccess) return e;
                    if (ned) hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, RF == 0 ? 2 : 1, true>), grid, block, lds, stream, p, about);
                    else hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, 1, true>), grid, block, lds, stream, p, about);
                    return hipGetLastError();
                }
                GINSIM_NAME_OR("ginsim::mc_kernel<%d, %d, false, true, %d, false>", RF, ALGOS, (ned ? 2 : 1) + (plain ? 2 : 0))
                if constexpr (kPlainForm) {
                    if (plain) {
                        if (ned) hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, 4>), grid, block, lds, stream, p, about);
                        else hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, 3>), grid, block, lds, stream, p, about);
                        return hipGetLastError();
                    }
                }
                if (hipError_t e = write_shift(false); e != hipSuccess) return e;
                if (ned) hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, RF == 0 ? 2 : 1>), grid, block, lds, stream, p, about);
                else hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, 1>), grid, block, lds, stream, p, about);
                return hipGetLastError();
            } else {                                // the simple model (every standard IMU grade), statistics in the state's own units
                GINSIM_NAME_OR("ginsim::mc_kernel<%d, %d, false, false, %d, false>", RF, ALGOS, plain ? 3 : 1)
                if constexpr (kPlainForm) {
                    if (plain) {
                        hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, false, 3>), grid, block, lds, stream, p, about);
                        return hipGetLastError();
                    }
                }
                if (hipError_t e = write_shift(false); e != hipSuccess) return e;
                hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, false, 1>), grid, block, lds, stream, p, about);
                return hipGetLastError();
            }
        }
    }
    if constexpr (WD) {
        if (any_vibration(p)) {
            GINSIM_NAME_OR("ginsim::mc_kernel<%d, %d, false, true, 0, true>", RF, ALGOS)
            hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, true, 0, true>), grid, block, lds, stream, p, about);
            return hipGetLastError();
        }
    }
    GINSIM_NAME_OR("ginsim::mc_kernel<%d, %d, false, %s, 0, false>", RF, ALGOS, tf(WD))
    hipLaunchKernelGGL((mc_kernel<RF, ALGOS, false, WD>), grid, block, lds, stream, p, about);
    return hipGetLastError();
}

template <int RF, int ALGOS>
static hipError_t launch2(const ginsim_mc_params& p, hipStream_t stream, char* name, size_t cap, double* about) {
    if (p.given_sensors) {
        if constexpr (ALGOS != 0) {
            GINSIM_NAME_OR("ginsim::mc_kernel<%d, %d, true, false, 0, false>", RF, ALGOS)
            const int tb = p.block_threads > 0 ? p.block_threads : kBlock;
            const int64_t waves = (p.runs + kWave - 1) / kWave;
            const int per_cu = waves <= 1024 ? 1 : 2;
            const size_t lds = p.block_threads > 0 ? 0 : kLdsPerCu / (per_cu + 1) + 1024;
            hipLaunchKernelGGL((mc_kernel<RF, ALGOS, true, false>), dim3((unsigned)((p.runs + tb - 1) / tb)), dim3(tb), lds, stream, p, about);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;        // given sensors without an algorithm: rejected by the C ABI
    }
    // the simple-model variant of the two-algorithm ref_frame 0 kernels is the one instantiation that spills: use the general one
    constexpr bool kSimpleFits = !(RF == 0 && ALGOS == (GINSIM_ALGO_FREE | GINSIM_ALGO_ODO));
    // online process statistics: the simple model where the launch has no NED record to keep (round 6: that instantiation no longer
    // spills -- 245 registers -- and saves the general model's 30 selects per step; with the NED record it would, 36 B per lane)
    const bool proc = p.out_proc[0] || p.out_proc[1];
    const bool general_ps = proc && getenv("GINSIM_PS_GENERAL") != nullptr;       // read per call: the tests compare the two kernels bit for bit
    if (any_white_drift(p) || any_vibration(p) || !kSimpleFits || (proc && p.ref_frame == 0 && p.proc_pos_ned) || general_ps)
        return launch3<RF, ALGOS, true>(p, stream, name, cap, about);
    if constexpr (kSimpleFits) return launch3<RF, ALGOS, false>(p, stream, name, cap, about);
    return hipErrorInvalidValue;
}

template <int RF>
static hipError_t launch1(const ginsim_mc_params& p, hipStream_t stream, char* name, size_t cap, double* about) {
    switch (p.algo_mask) {
        case 0: return launch2<RF, 0>(p, stream, name, cap, about);      // sensors only (Sim without an algorithm)
        case GINSIM_ALGO_FREE: return launch2<RF, GINSIM_ALGO_FREE>(p, stream, name, cap, about);
        case GINSIM_ALGO_ODO: return launch2<RF, GINSIM_ALGO_ODO>(p, stream, name, cap, about);
        default: return launch2<RF, GINSIM_ALGO_FREE | GINSIM_ALGO_ODO>(p, stream, name, cap, about);
    }
}

hipError_t launch_mc(const ginsim_mc_params& p, hipStream_t stream, char* name, size_t cap, double* about) {
    return p.ref_frame == 1 ? launch1<1>(p, stream, name, cap, about) : launch1<0>(p, stream, name, cap, about);
}

// ---------------------------------------------------------------------------------------------------
// Sensor series for FEW runs (Sim.run(1) as a data generator, the Allan flow of BASELINE config 5): with one lane per
// run the time loop of mc_kernel is a single sequential chain (n = 1 440 000 samples -> 2.8 s on one lane), and there are
// no runs to fill the chip with.  Here the TIME axis is the parallel one: a lane holds TWO CONSECUTIVE samples (kSpan), a
// wavefront covers 128 consecutive samples of one run per step and walks a chunk of L samples, so
//   * the Philox counter (sample, block, run) makes the twelve normals of a sample a per-lane computation;
//   * a lane writes 16 contiguous bytes of ONE series per step, a wavefront 1 KB -- in the series-major layout
//     [run][axis][n] (sensor_layout 1), which is what ginsim_allan reads: the Allan flow needs no re-layout;
//   * the one sequential thing, the Gauss-Markov recurrence d[j+1] = a d[j] + b w[j] (pathgen.py:583-590), is linear:
//     inside a lane it is evaluated as written, across the lanes of a step it is a weighted inclusive scan of the lanes'
//     two-sample sums with ratio a^2 (Hillis-Steele in DPP: row_shr 1/2/4/8 with the wave-uniform weights a^2, a^4, a^8,
//     a^16, then row_bcast:15 / :31 with the per-lane weights), ONE scan per 128 samples and axis (round 4 had a lane = a
//     sample and scanned every 64: 166 of the 560 vector instructions of a step went into it; the kernels are bound by
//     instruction issue, 1.24 ns per instruction and wave-step in either form), the carry of the previous 128 samples enters
//     at lane 0, and across chunks it is three launches:
//       pass A  chunk-end value of every chunk integrated from zero (drift normals only; a lane accumulates its
//               steps with weight a^128, one weighted wave reduction at the end of the chunk)
//       pass S  chunk-end values -> chunk-START values, start[k+1] = a^L start[k] + end[k]: the same scan, one
//               wavefront per (run, axis)
//       pass B  regenerates the normals (counter-based RNG: no state to carry) and emits
//               truth + bias + drift + white (pathgen.py:500, 562) with the recurrence started from start[k].
// Same normals and the same recurrence as the lane-per-run kernels; only the association of its sums differs (powers of
// a instead of repeated multiplication: a relative 1e-16 on a drift of ~1e-5, far below the 1e-12 / 1e-14 sensor
// tolerances and inside "an ulp of the terms" of the emitted sums, tests/test_gpu_edge_cases.py).
// Round 3's version gave a THREAD a chunk of up to 4096 samples: 44 workgroups for config 5's 32 x 1 440 000 samples
// (17 % of the chip), run-fastest stores 256 B apart, 6.0 ms + a 1.0 ms re-layout in front of a 0.5 ms Allan call.
// wave-uniform weights of the weighted scan with ratio q (ScanWeights below)
struct ScanQ { double q1, q2, q4, q8, q16; };
typedef const ScanQ __attribute__((address_space(4))) * scanq_ptr;

struct SeriesPlan {
    double* carry;          // [runs][nchunks][6]: pass A chunk-end values, pass S overwrites them with chunk-start values
    double a_pow[6];        // gm_a ^ L for accel xyz, gyro xyz
    int64_t nchunks;
    int64_t sr, sc;         // element (run r, axis c, sample j) of a 3-axis sensor lives at r sr + c sc + j
    int64_t odo_sr;         // and of the odometer at r odo_sr + j
    int32_t L;              // samples per chunk, a multiple of 64 kSpan (one step of a wavefront)
    int32_t pad;
    ScanQ   qs[6];          // powers of gm_a ^ kSpan (the scan over the lanes of a step: a lane holds kSpan consecutive samples)
    ScanQ   qL[6];          // powers of gm_a ^ L (pass S)
    double  a_step[6];      // gm_a ^ (64 kSpan) (pass A: the same lane, one step later)
};
typedef const SeriesPlan __attribute__((address_space(4))) * plan_ptr;
// the plan is the second kernel argument: it follows the parameter block in the kernarg segment
__device__ __forceinline__ plan_ptr kernarg_plan(size_t offset) {
    auto p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (plan_ptr)(p + offset);
}

// Lanes the control cannot serve (the first lanes of a row for row_shr, row 0 for row_bcast) read 0.0 (bound_ctrl); every
// row is written (row_mask 0xf), so the instruction needs no defined previous value of its destination -- with a row mask
// the compiler had to zero the destination first: 84 v_mov per step.  The rows a broadcast must not reach get weight 0.
template <int CTRL>
__device__ __forceinline__ double dpp_or_zero(double x) {
    const int xl = __double2loint(x), xh = __double2hiint(x);
    const int lo = __builtin_amdgcn_update_dpp(xl, xl, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(xh, xh, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// lane l <- lane l-1, lane 0 <- first
__device__ __forceinline__ double wave_shift_up(double x, double first) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(x), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_lane63(double x) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// Weights of the weighted scan with ratio q: wave-uniform q, q^2, q^4, q^8, q^16 (host-computed, read from the kernarg
// segment where they are used: SGPR pairs, re-loaded per use instead of 48 VGPRs) and, per lane, q^((l & 15) + 1),
// q^((l & 31) + 1) -- zero in the rows that broadcast does not feed.

struct ScanWeights {
    double w15, w31;
    __device__ __forceinline__ void init(scanq_ptr q, int lane) {
        const int p = (lane & 15) + 1;                  // 1 .. 16
        double w = (p & 1) ? q->q1 : 1.0;
        w = (p & 2) ? w * q->q2 : w;
        w = (p & 4) ? w * q->q4 : w;
        w = (p & 8) ? w * q->q8 : w;
        w = (p & 16) ? q->q16 : w;
        w15 = (lane & 16) ? w : 0.0;                                    // rows 1 and 3 take the row before them
        w31 = (lane & 32) ? ((lane & 16) ? w * q->q16 : w) : 0.0;       // rows 2 and 3 take lane 31
    }
    // e[l] = sum_{i <= l} q^(l - i) u[i]
    __device__ __forceinline__ double inclusive(double e, scanq_ptr q) const {
        e = __builtin_fma(q->q1, dpp_or_zero<0x111>(e), e);        // row_shr:1
        e = __builtin_fma(q->q2, dpp_or_zero<0x112>(e), e);        // row_shr:2
        e = __builtin_fma(q->q4, dpp_or_zero<0x114>(e), e);        // row_shr:4
        e = __builtin_fma(q->q8, dpp_or_zero<0x118>(e), e);        // row_shr:8   -> scans of the four rows of 16
        e = __builtin_fma(w15, dpp_or_zero<0x142>(e), e);          // row_bcast:15: lane 15 of a row to the next row
        e = __builtin_fma(w31, dpp_or_zero<0x143>(e), e);          // row_bcast:31: lane 31 to rows 2 and 3
        return e;
    }
};

static void scanq_host(double q, ScanQ* out) {
    out->q1 = q; out->q2 = q * q; out->q4 = out->q2 * out->q2; out->q8 = out->q4 * out->q4; out->q16 = out->q8 * out->q8;
}

constexpr int kSeriesBlock = 256;       // four wavefronts = four chunks per workgroup
// Samples per lane: measured on config 5 (32 x 1 440 000; pass B alone, rocprofv3): 1 -> 601 us, 2 -> 516 us, 4 -> 673 us.  Four halve
// the scan's share again (335 vector instructions per 64 samples against 410) but need 168 registers (three wavefronts per
// SIMD) and read the truth rows with 96-byte lane strides: 48 cache lines per load instruction.
constexpr int kSpan = 2;                // consecutive samples of a lane
constexpr int kSeriesWaves = 4;         // wavefronts per SIMD the register allocator is held to (128 registers; the vibration
                                        // variant, with its sines: three)
constexpr int kGroup = 64 * kSpan;      // samples of one step of a wavefront

// the consecutive doubles of a lane in one series: 16-byte streaming stores (a series starts on an 8-byte boundary only)
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef f64x2 f64x2_a8 __attribute__((aligned(8)));
__device__ __forceinline__ void st_span(double* p, const double (&v)[kSpan]) {
    if (kSpan == 1) st(p, v[0]);
#pragma unroll
    for (int i = 0; i + 1 < kSpan; i += 2) __builtin_nontemporal_store(f64x2{v[i], v[i + 1]}, reinterpret_cast<f64x2_a8*>(p + i));
}

// PASS 0: pass A; 1: pass B; 2: pass B with the vibration term of Sim(env=...) (a per-sample term: nothing to scan); 3: pass B for
// the simple sensor model (no white-drift axis, no constant bias -- every standard IMU grade of imu_model.py: the six wave-uniform
// selects and the bias additions are compiled out, x + 0.0 == x)
template <int PASS>
__global__ void __launch_bounds__(kSeriesBlock, PASS == 2 ? kSeriesWaves - 1 : kSeriesWaves) series_kernel(const ginsim_mc_params a, const SeriesPlan pl) {
    constexpr bool VIB = PASS == 2;
    constexpr bool WD = PASS != 3;
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    // grid: x = groups of four chunks, y = run; everything below is wave-uniform (SGPRs) except `lane`
    const int64_t c = (int64_t)blockIdx.x * (kSeriesBlock / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = blockIdx.y;
    if (c >= pl.nchunks) return;
    const int64_t j0 = c * pl.L, j1 = (j0 + pl.L < a.n) ? j0 + pl.L : a.n;
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    double* cb = pl.carry + (r * pl.nchunks + c) * 6;
    const params_ptr kp = kernarg_params();
    double ga[6], gb[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ga[k] = kp->accel.gm_a[k]; gb[k] = kp->accel.gm_b[k];
        ga[3 + k] = kp->gyro.gm_a[k]; gb[3 + k] = kp->gyro.gm_b[k];
    }
    const plan_ptr kq = kernarg_plan(sizeof(ginsim_mc_params));
    if (PASS == 0) {
        // chunk-end value from zero: a lane folds its samples of every step (weight a), its steps with weight a^(64 kSpan),
        // then one scan over the lanes
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto fold = [&](const int64_t jg, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;        // every sample of the step inside the series: no masks
            double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < kSpan; ++i) {
                const int64_t j = jg + kSpan * lane + i;
                const bool on = FULL || j < j1;
                double z0[6], z1[6];
                normal_pairs<S_ACC_D_XY, 6>(key, (uint32_t)(on ? j : j1 - 1), z0, z1, tab);
                const double zd[6] = {z0[0], z1[0], z0[1], z0[3], z1[3], z0[4]};
#pragma unroll
                for (int k = 0; k < 6; ++k) e[k] = __builtin_fma(ga[k], e[k], on ? gb[k] * zd[k] : 0.0);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] = __builtin_fma(kernarg_plan(sizeof(ginsim_mc_params))->a_step[k], acc[k], e[k]);
        };
        int64_t jg = j0;
        for (; jg + kGroup <= j1; jg += kGroup) fold(jg, std::true_type{});
        if (jg < j1) fold(jg, std::false_type{});
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ScanWeights sw;                     // of the scan over the LANES: ratio a^kSpan
            sw.init(&kq->qs[k], lane);
            const double e = sw.inclusive(acc[k], &kq->qs[k]);
            if (lane == 63) cb[k] = e;
        }
        return;
    }
    ScanWeights sw[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sw[k].init(&kq->qs[k], lane);

    // ---- pass B
    const model_ptr ma = &kp->accel, mg = &kp->gyro;
    double carry[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) carry[k] = as_uniform(cb)[k];
    double* const oa = a.out_accel ? a.out_accel + r * pl.sr : nullptr;
    double* const og = a.out_gyro ? a.out_gyro + r * pl.sr : nullptr;
    double* const oo = a.out_odo ? a.out_odo + r * pl.odo_sr : nullptr;
    Vec3 vpa{0.0, 0.0, 0.0}, vpg{0.0, 0.0, 0.0};
    if (VIB) {
        vpa = vibration_phase<S_ACC_VIB_PHASE>(&kp->vib_accel, key);
        vpg = vibration_phase<S_GYR_VIB_PHASE>(&kp->vib_gyro, key);
    }
    // one step = 64 kSpan consecutive samples; FULL: all of them inside the series (every step but the last of a ragged series).
    // The words of the three Philox blocks of the lane's samples first (12 registers each), then one sensor after the other:
    // transform, recurrence, sums, stores -- the scheduling barriers keep the two sensors' working sets apart.
    auto step = [&](const int64_t jg, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t jl = jg + kSpan * lane;
        bool on[kSpan];
        uint32_t wa[kSpan][6], wb[kSpan][6];
#pragma unroll
        for (int i = 0; i < kSpan; ++i) {
            const int64_t j = jl + i;
            on[i] = FULL || j < j1;
            draw_streams<S_ACC_D_XY, 6>(key, (uint32_t)(on[i] ? j : j1 - 1), wa[i], wb[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        auto sensor = [&](auto sensor_tag) {
            constexpr int S = decltype(sensor_tag)::value;          // 0: accelerometer (streams 0..2), 1: gyroscope (3..5)
            const model_ptr m = S ? mg : ma;
            const double* const truth = S ? a.ref_gyro : a.ref_accel;
            double* const out = S ? og : oa;
            // the recurrence d[j+1] = a d[j] + b w[j]: inside a lane as written, across the lanes the weighted scan of the
            // lanes' sums (ratio a^kSpan); the drift at the step's first sample enters at lane 0
            double u[kSpan][3], d[kSpan][3], o[kSpan][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int K = 3 * S + k;
#pragma unroll
                for (int i = 0; i < kSpan; ++i) {
                    const uint32_t w = k == 0 ? wa[i][3 * S] : (k == 1 ? wb[i][3 * S] : wa[i][3 * S + 1]);
                    u[i][k] = on[i] ? gb[K] * (double)normal_icdf(w, tab) : 0.0;
                }
                double e = u[0][k];
#pragma unroll
                for (int i = 1; i < kSpan; ++i) e = __builtin_fma(ga[K], e, u[i][k]);
                e = lane == 0 ? __builtin_fma(kernarg_plan(sizeof(ginsim_mc_params))->qs[K].q1, carry[K], e) : e;
                const double inc = sw[K].inclusive(e, &kernarg_plan(sizeof(ginsim_mc_params))->qs[K]);   // drift at the next lane's first sample
                d[0][k] = wave_shift_up(inc, carry[K]);                                                    // at this lane's
                carry[K] = wave_lane63(inc);
#pragma unroll
                for (int i = 1; i < kSpan; ++i) d[i][k] = __builtin_fma(ga[K], d[i - 1][k], u[i - 1][k]);
            }
            // the sums of sense3 (pathgen.py:500, 562), same order of operations
#pragma unroll
            for (int i = 0; i < kSpan; ++i) {
                if (FULL || on[i]) {
                    const int64_t j = jl + i;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t w = k == 0 ? wb[i][3 * S + 1] : (k == 1 ? wa[i][3 * S + 2] : wb[i][3 * S + 2]);
                        const double ud = (WD && m->white_drift[k]) ? u[i][k] : d[i][k];
                        const double tb = WD ? truth[3 * j + k] + m->bias[k] : truth[3 * j + k];
                        o[i][k] = tb + ud + m->white[k] * (double)normal_icdf(w, tab);
                    }
                    if (VIB) {          // added last, as pathgen.py:500, 562 do
                        const Vec3 v = S ? add_vibration<S_GYR_VIB_XY>(Vec3{o[i][0], o[i][1], o[i][2]}, &kernarg_params()->vib_gyro, key, (uint32_t)j, tab, vpg, Vec3{0.0, 0.0, 0.0})
                                         : add_vibration<S_ACC_VIB_XY>(Vec3{o[i][0], o[i][1], o[i][2]}, &kernarg_params()->vib_accel, key, (uint32_t)j, tab, vpa, Vec3{0.0, 0.0, 0.0});
                        o[i][0] = v.x; o[i][1] = v.y; o[i][2] = v.z;
                    }
                    if (!FULL && out) { st(out + j, o[i][0]); st(out + pl.sc + j, o[i][1]); st(out + 2 * pl.sc + j, o[i][2]); }
                }
            }
            if (FULL && out) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double v[kSpan];
#pragma unroll
                    for (int i = 0; i < kSpan; ++i) v[i] = o[i][k];
                    st_span(out + k * pl.sc + jl, v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        sensor(std::integral_constant<int, 0>{});
        sensor(std::integral_constant<int, 1>{});
        if (oo) {
            double od[kSpan];
#pragma unroll
            for (int i = 0; i < kSpan; ++i) {
                if (FULL || on[i]) {
                    const int64_t j = jl + i;
                    double y0, y1;
                    normal_pair(key, S_ODO, (uint32_t)j, y0, y1, tab);
                    od[i] = kp->odo_scale * a.ref_odo[j] + kp->odo_stdv * y0;     // pathgen.py:639-640
                    if (!FULL) st(oo + j, od[i]);
                }
            }
            if (FULL) st_span(oo + jl, od);
        }
    };
    int64_t jg = j0;
    for (; jg + kGroup <= j1; jg += kGroup) step(jg, std::true_type{});
    if (jg < j1) step(jg, std::false_type{});
}

// chunk-end values -> chunk-start values, one wavefront per (run, axis): start[0] = 0, start[k+1] = a^L start[k] + end[k]
// is the weighted scan again (ratio a^L), 64 chunks per step
__global__ void __launch_bounds__(64) series_scan_kernel(const SeriesPlan pl, int64_t runs) {
    const int lane = threadIdx.x;
    const int64_t id = blockIdx.x;
    if (id >= runs * 6) return;
    const int64_t r = id / 6;
    const int k = (int)(id % 6);
    double* cb = pl.carry + r * pl.nchunks * 6 + k;
    const scanq_ptr q = &kernarg_plan(0)->qL[k];
    ScanWeights sw;
    sw.init(q, lane);
    double carry = 0.0;                                  // start value of chunk cbase
    for (int64_t cbase = 0; cbase < pl.nchunks; cbase += 64) {
        const int64_t c = cbase + lane;
        const bool on = c < pl.nchunks;
        double u = on ? cb[c * 6] : 0.0;
        u = lane == 0 ? __builtin_fma(pl.a_pow[k], carry, u) : u;
        const double e = sw.inclusive(u, q);             // start of chunk c + 1
        const double s = wave_shift_up(e, carry);        // start of chunk c
        carry = wave_lane63(e);
        if (on) cb[c * 6] = s;
    }
}

// which pass B a launch takes (ginsim_mc_kernel_name reports it)
int series_pass_b(const ginsim_mc_params& p) { return any_vibration(p) ? 2 : (any_white_drift(p) ? 1 : 3); }

// sensors only, few runs, long series
bool series_path_applies(const ginsim_mc_params& p) {
    return p.algo_mask == 0 && !p.given_sensors && p.precision == 0 && !p.wave_trace && p.block_threads == 0 && !any_psd_vibration(p) &&
           p.runs <= 1024 && p.n >= 2048 && (p.sensor_layout == 1 || p.runs == 1);
}

int64_t series_chunks(const ginsim_mc_params& p, int32_t* L_out) {
    // ~16 384 wavefronts over the chip (1024 SIMDs, several rounds of a few wavefronts each), chunks of 256 .. 8192 samples in
    // whole wave-steps, and at most 1024 chunks per run where that fits (pass S walks them 64 at a time)
    int64_t L = (p.n * p.runs + 16383) / 16384;
    const int64_t lmin = (p.n + 1023) / 1024;
    if (L < lmin) L = lmin;
    if (L < 256) L = 256;
    if (L > 8192) L = 8192;
    L = (L + kGroup - 1) / kGroup * kGroup;
    *L_out = (int32_t)L;
    return (p.n + L - 1) / L;
}

hipError_t launch_series(const ginsim_mc_params& p, double* carry, hipStream_t stream) {
    SeriesPlan pl;
    pl.carry = carry;
    pl.nchunks = series_chunks(p, &pl.L);
    for (int k = 0; k < 6; ++k) {
        const double aa = k < 3 ? p.accel.gm_a[k] : p.gyro.gm_a[k - 3];
        double v = 1.0;
        for (int i = 0; i < pl.L; ++i) v *= aa;
        pl.a_pow[k] = v;
        static_assert(kSpan == 2, "the host's powers of gm_a");
        scanq_host(aa * aa, &pl.qs[k]);
        scanq_host(v, &pl.qL[k]);
        const double a16s = pl.qs[k].q16, a32s = a16s * a16s;     // gm_a ^ (16 span), ^ (32 span)
        pl.a_step[k] = a32s * a32s;                                  // gm_a ^ (64 span): one step of a wavefront
    }
    pl.pad = 0;
    // the sample index is the contiguous one in both layouts the path serves (series_path_applies: layout 1, or one run)
    if (p.sensor_layout == 1) { pl.sr = 3 * p.n; pl.sc = p.n; pl.odo_sr = p.n; }
    else { pl.sr = 0; pl.sc = p.n; pl.odo_sr = 0; }
    const dim3 grid((unsigned)((pl.nchunks + kSeriesBlock / 64 - 1) / (kSeriesBlock / 64)), (unsigned)p.runs), block(kSeriesBlock);
    hipLaunchKernelGGL((series_kernel<0>), grid, block, 0, stream, p, pl);
    hipLaunchKernelGGL(series_scan_kernel, dim3((unsigned)(p.runs * 6)), dim3(64), 0, stream, pl, p.runs);
    if (any_vibration(p)) hipLaunchKernelGGL((series_kernel<2>), grid, block, 0, stream, p, pl);
    else if (any_white_drift(p)) hipLaunchKernelGGL((series_kernel<1>), grid, block, 0, stream, p, pl);
    else hipLaunchKernelGGL((series_kernel<3>), grid, block, 0, stream, p, pl);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Auxiliary sensors: one thread per (sample, run), run fastest.  gps_gen: pathgen.py:621-624; mag_gen: :658-661.
__global__ void __launch_bounds__(256) aux_gps_kernel(const ginsim_aux_params a) {
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();

    MathConsts mk;
    mk.init<false>();
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.m * a.runs) return;
    const int64_t r = idx % a.runs, k = idx / a.runs;
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    double z0[3], z1[3];
    normal_pairs<S_GPS_P_XY, 3>(key, (uint32_t)k, z0, z1, tab);
    const double z[6] = {z0[0], z1[0], z0[1], z1[1], z0[2], z1[2]};     // pos x,y,z  vel x,y,z
    const int64_t plane = a.m * a.runs;
#pragma unroll
    for (int c = 0; c < 6; ++c) a.out_gps[c * plane + idx] = a.ref_gps[6 * k + c] + a.gps_sigma[c] * z[c];
}

__global__ void __launch_bounds__(256) aux_mag_kernel(const ginsim_aux_params a) {
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();

    MathConsts mk;
    mk.init<false>();
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n * a.runs) return;
    const int64_t r = idx % a.runs, j = idx / a.runs;
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};
    double z0[2], z1[2];
    normal_pairs<S_MAG_XY, 2>(key, (uint32_t)j, z0, z1, tab);
    const double z[3] = {z0[0], z1[0], z0[1]};
    const double v[3] = {a.ref_mag[3 * j] + a.mag_hi[0], a.ref_mag[3 * j + 1] + a.mag_hi[1], a.ref_mag[3 * j + 2] + a.mag_hi[2]};
    const int64_t plane = a.n * a.runs;
#pragma unroll
    for (int c = 0; c < 3; ++c)     // (ref + hi) . si^T  + std * N
        a.out_mag[c * plane + idx] = a.mag_si[3 * c] * v[0] + a.mag_si[3 * c + 1] * v[1] + a.mag_si[3 * c + 2] * v[2] + a.mag_std[c] * z[c];
}

hipError_t launch_aux(const ginsim_aux_params& p, hipStream_t s) {
    if (p.out_gps && p.ref_gps && p.m > 0)
        hipLaunchKernelGGL(aux_gps_kernel, dim3((unsigned)((p.m * p.runs + 255) / 256)), dim3(256), 0, s, p);
    if (p.out_mag && p.ref_mag && p.n > 0)
        hipLaunchKernelGGL(aux_mag_kernel, dim3((unsigned)((p.n * p.runs + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// RNG self-test: normals (and raw Philox words) of one (seed, run, stream), sample index = global lane.
__global__ void rng_probe_kernel(uint64_t seed, uint64_t run, uint32_t stream, int64_t count,
                                 double* __restrict__ z0, double* __restrict__ z1, uint32_t* __restrict__ words) {
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();

    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)run, (uint32_t)(run >> 32)};
    double a, b;
    normal_pair(key, stream, (uint32_t)j, a, b, tab);
    z0[j] = a;
    z1[j] = b;
    if (words) {
        const u32x4 w = philox4x32((uint32_t)j, stream, key.r0, key.r1, key.k0, key.k1);      // raw block (j, stream)
        words[4 * j + 0] = w.x; words[4 * j + 1] = w.y; words[4 * j + 2] = w.z; words[4 * j + 3] = w.w;
    }
}

hipError_t launch_rng_probe(uint64_t seed, uint64_t run, uint32_t stream, int64_t count, double* z0, double* z1,
                            uint32_t* words, hipStream_t stream_h) {
    const int tb = 256;
    hipLaunchKernelGGL(rng_probe_kernel, dim3((unsigned)((count + tb - 1) / tb)), dim3(tb), 0, stream_h, seed, run,
                       stream, count, z0, z1, words);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Layout helpers for the host-buffer boundary: [R][n][C] (reference per-run arrays) <-> [C][n][R].
__global__ void aos_to_soa_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t R, int64_t n, int C) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over n*R, run fastest
    if (idx >= n * R) return;
    const int64_t r = idx % R, j = idx / R;
    for (int c = 0; c < C; ++c) dst[(c * n + j) * R + r] = src[(r * n + j) * C + c];
}

// gather selected runs: series [C][n][runs] -> out [nsel][n][C]
__global__ void gather_runs_kernel(const double* __restrict__ series, int C, int64_t n, int64_t runs,
                                   const int64_t* __restrict__ ids, int nsel, double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over nsel*n*C, component fastest
    const int64_t total = (int64_t)nsel * n * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t j = (idx / C) % n;
    const int64_t k = idx / (C * n);
    out[idx] = series[((int64_t)c * n + j) * runs + ids[k]];
}

// the same from a series-major buffer [runs][C][n] (sensor_layout 1): out [nsel][n][C]
__global__ void gather_series_kernel(const double* __restrict__ series, int C, int64_t n, const int64_t* __restrict__ ids, int nsel,
                                     double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over nsel*C*n, sample fastest (the reads coalesce)
    const int64_t total = (int64_t)nsel * n * C;
    if (idx >= total) return;
    const int64_t j = idx % n;
    const int c = (int)((idx / n) % C);
    const int64_t k = idx / (n * C);
    out[(k * n + j) * C + c] = series[(ids[k] * C + c) * n + j];
}

hipError_t launch_gather_series(const double* series, int C, int64_t n, const int64_t* ids, int nsel, double* out, hipStream_t s) {
    const int tb = 256;
    const int64_t total = (int64_t)nsel * n * C;
    hipLaunchKernelGGL(gather_series_kernel, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, s, series, C, n, ids, nsel, out);
    return hipGetLastError();
}

// The normal transform on given words (test hook): words 0-1 are taken as one half block -- z0 from word 0, z1 from
// word 1 (words 2-3 unused).
__global__ void normal_transform_kernel(const uint32_t* __restrict__ words, int64_t count, double* __restrict__ z0, double* __restrict__ z1) {
    __shared__ uint32_t ntab[kNormalLdsWords];
    const NormalTables tab = fill_normal_tables(ntab, threadIdx.x, blockDim.x);
    __syncthreads();

    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t ra[1] = {words[4 * i]}, ang[1] = {words[4 * i + 1]};
    float a[1], b[1];
    normal_transform<1>(ra, ang, a, b, tab);
    z0[i] = (double)a[0];
    z1[i] = (double)b[0];
}

hipError_t launch_normal_transform(const uint32_t* words, int64_t count, double* z0, double* z1, hipStream_t s) {
    hipLaunchKernelGGL(normal_transform_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, words, count, z0, z1);
    return hipGetLastError();
}

// [C][n][R] -> [R][C][n]: per component a (n x R) -> (R x n) transpose through a padded LDS tile of 64 samples x up to 64 runs.
// The tile's source rows are read as ONE flat range when the tile spans whole rows (R <= 64: 64 x R contiguous doubles, every
// lane busy whatever R is -- with a lane per run, 32 runs left half of every wavefront idle and the re-layout of config 5's
// 2 x 1.1 GB ran at 2.6 TB/s; now 4.6); the writes are 512-byte rows of 64 samples, one per run of the tile.  (Tiles of 128
// samples for few runs -- 1 KiB rows on the write side -- measured no faster.)
__global__ void __launch_bounds__(256) runs_to_series_kernel(const double* __restrict__ in, double* __restrict__ out, int C,
                                                            int64_t n, int64_t R) {
    __shared__ double tile[64][65];
    const int c = blockIdx.z;
    const int64_t j0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
    const int rt = (int)(R - r0 < 64 ? R - r0 : 64);            // runs in this tile
    const int jt = (int)(n - j0 < 64 ? n - j0 : 64);            // samples in this tile
    const double* src = in + (int64_t)c * n * R + j0 * R + r0;
    const bool pow2 = (rt & (rt - 1)) == 0;
    const int sh = 31 - __builtin_clz(rt);
    for (int e = threadIdx.x; e < jt * rt; e += 256) {
        const int j = pow2 ? (e >> sh) : e / rt;
        const int r = e - j * rt;
        tile[j][r] = __builtin_nontemporal_load(&src[(int64_t)j * R + r]);
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (tx < jt)
        for (int k = ty; k < rt; k += 4) __builtin_nontemporal_store(tile[tx][k], &out[((r0 + k) * C + c) * n + j0 + tx]);
}

hipError_t launch_runs_to_series(const double* in, double* out, int C, int64_t n, int64_t R, hipStream_t s) {
    hipLaunchKernelGGL(runs_to_series_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((R + 63) / 64), (unsigned)C), dim3(256), 0, s,
                       in, out, C, n, R);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(const double* src, double* dst, int64_t R, int64_t n, int C, hipStream_t s) {
    const int tb = 256;
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((unsigned)((n * R + tb - 1) / tb)), dim3(tb), 0, s, src, dst, R, n, C);
    return hipGetLastError();
}

hipError_t launch_gather_runs(const double* series, int C, int64_t n, int64_t runs, const int64_t* ids, int nsel,
                              double* out, hipStream_t s) {
    const int tb = 256;
    const int64_t total = (int64_t)nsel * n * C;
    hipLaunchKernelGGL(gather_runs_kernel, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, s, series, C, n, runs,
                       ids, nsel, out);
    return hipGetLastError();
}

}  // namespace ginsim

