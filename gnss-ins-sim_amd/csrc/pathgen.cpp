// Truth generator: motion commands -> true position / velocity / attitude and ideal IMU, GPS, odometer
// signals.  Native restatement of pathgen.path_gen with sim_osr == 1 (the only value Sim ever passes,
// gnss_ins_sim/sim/ins_sim.py:451):
//
//   path_gen                 gnss_ins_sim/pathgen/pathgen.py:26-329
//   calc_true_sensor_output  gnss_ins_sim/pathgen/pathgen.py:331-411
//   parse_motion_def         gnss_ins_sim/pathgen/pathgen.py:413-439
//
// It is a single sequential recurrence (no Monte-Carlo axis) that runs once per Sim.run, so it runs on
// the host inside libginsim.so and its output is uploaded once; every MC lane then reads it through the
// scalar cache.  This translation unit is compiled with -ffp-contract=off: the segment-completion test
// (pathgen.py:221-223) compares against a threshold, and the sample count n must not depend on whether a
// compiler fused a multiply-add that NumPy evaluates as two roundings.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "ginsim.h"

namespace ginsim {
void set_error(const char* fmt, ...);

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kRe = 6378137.0;
constexpr double kFlat = 1.0 / 298.257223563;
constexpr double kEcc = 0.0818191908426215;
constexpr double kEsq = kEcc * kEcc;
constexpr double kWie = 7292115e-11;

struct V3 {
    double v[3];
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};

struct M3 { double m[3][3]; };      // body -> nav (c_nb of the reference)

inline V3 mul(const M3& a, const V3& x) {       // a . x
    V3 o;
    for (int i = 0; i < 3; ++i) o[i] = a.m[i][0] * x[0] + a.m[i][1] * x[1] + a.m[i][2] * x[2];
    return o;
}
inline V3 mul_t(const M3& a, const V3& x) {     // a^T . x
    V3 o;
    for (int i = 0; i < 3; ++i) o[i] = a.m[0][i] * x[0] + a.m[1][i] * x[1] + a.m[2][i] * x[2];
    return o;
}
inline V3 cross(const V3& a, const V3& b) {
    return V3{{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
inline double norm(const V3& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline double clamp(double x, double lim) { return x > lim ? lim : (x < -lim ? -lim : x); }

// transpose of attitude.euler2dcm(att,'zyx') (attitude.py:344-371)
inline M3 body_to_nav(const V3& att) {
    const double cy = std::cos(att[0]), cp = std::cos(att[1]), cr = std::cos(att[2]);
    const double sy = std::sin(att[0]), sp = std::sin(att[1]), sr = std::sin(att[2]);
    M3 c;
    c.m[0][0] = cp * cy;  c.m[1][0] = cp * sy;  c.m[2][0] = -sp;
    c.m[0][1] = sr * sp * cy - cr * sy;  c.m[1][1] = sr * sp * sy + cr * cy;  c.m[2][1] = cp * sr;
    c.m[0][2] = sp * cr * cy + sy * sr;  c.m[1][2] = sp * cr * sy - cy * sr;  c.m[2][2] = cp * cr;
    return c;
}

struct Earth { double rm, rn, g, sl, cl; };
inline Earth earth(double lat, double h) {      // geoparams.geo_param, geoparams.py:25-53
    Earth e;
    e.sl = std::sin(lat);
    e.cl = std::cos(lat);
    const double s2 = e.sl * e.sl;
    e.rm = (kRe * (1 - kEsq)) / (std::sqrt(1.0 - kEsq * s2) * (1.0 - kEsq * s2));
    e.rn = kRe / std::sqrt(1.0 - kEsq * s2);
    const double g1 = 9.7803253359 * (1 + 0.00193185265241 * s2) / std::sqrt(1.0 - kEsq * s2);
    e.g = g1 * (1.0 - (2.0 / kRe) * (1.0 + kFlat + 0.00344978650684 - 2.0 * kFlat * s2) * h + 3.0 * h * h / kRe / kRe);
    return e;
}

// Python float % for a positive divisor
inline double py_mod(double x, double m) {
    double r = std::fmod(x, m);
    if (r != 0.0 && r < 0.0) r += m;
    return r;
}
inline double wrap_pi(double x) {               // attitude.angle_range_pi, attitude.py:799-812
    x = py_mod(x, 2.0 * kPi);
    return x > kPi ? x - 2.0 * kPi : x;
}
// attitude.euler_angle_range_three_axis, attitude.py:772-797
inline V3 euler_range(const V3& a) {
    double a1 = a[0], a2 = wrap_pi(a[1]), a3 = a[2];
    if (a2 > 0.5 * kPi) { a2 = kPi - a2; a1 += kPi; a3 += kPi; }
    else if (a2 < -0.5 * kPi) { a2 = -kPi - a2; a1 += kPi; a3 += kPi; }
    return V3{{wrap_pi(a1), a2, wrap_pi(a3)}};
}

// pathgen.calc_true_sensor_output (pathgen.py:331-411): ideal accelerometer / gyroscope output and the position rate of one
// kinematic state.  g is used in ref_frame 1 only (ref_frame 0 takes the WGS-84 value at pos).  vel_dot_n (pathgen.py:391) is
// not used by path_gen itself; the C entry point below returns it as the reference function does.
struct TrueSensor { V3 acc, gyro, vel_dot_n, pos_dot; };
inline TrueSensor true_sensor_output(const V3& pos, const V3& vel_b, const V3& att, const M3& c_nb, const V3& vel_dot_b,
                                     const V3& att_dot, int ref_frame, double g, bool want_vel_dot_n) {
    TrueSensor o;
    const V3 vn = mul(c_nb, vel_b);
    V3 w_en{{0, 0, 0}}, w_ie{{0, 0, 0}};
    if (ref_frame == 0) {
        const Earth e = earth(pos[0], pos[2]);
        const double rm_e = e.rm + pos[2], rn_e = e.rn + pos[2];
        g = e.g;
        w_en = V3{{vn[1] / rn_e, -vn[0] / rm_e, -vn[1] * e.sl / e.cl / rn_e}};
        w_ie = V3{{kWie * e.cl, 0.0, -kWie * e.sl}};
        o.pos_dot = V3{{vn[0] / rm_e, vn[1] / rn_e / e.cl, -vn[2]}};
    } else {
        o.pos_dot = vn;
    }
    const double sh = std::sin(att[0]), ch = std::cos(att[0]);
    const V3 w_nb{{-sh * att_dot[1] + c_nb.m[0][0] * att_dot[2], ch * att_dot[1] + c_nb.m[1][0] * att_dot[2],
                   att_dot[0] + c_nb.m[2][0] * att_dot[2]}};
    o.vel_dot_n = V3{{0, 0, 0}};
    if (want_vel_dot_n) {
        const V3 a = mul(c_nb, vel_dot_b), c = cross(w_nb, vn);
        o.vel_dot_n = V3{{a[0] + c[0], a[1] + c[1], a[2] + c[2]}};
    }
    o.gyro = mul_t(c_nb, V3{{w_nb[0] + w_en[0] + w_ie[0], w_nb[1] + w_en[1] + w_ie[1], w_nb[2] + w_en[2] + w_ie[2]}});
    const V3 w_ie_b = mul_t(c_nb, w_ie);
    const V3 cor = cross(V3{{w_ie_b[0] + o.gyro[0], w_ie_b[1] + o.gyro[1], w_ie_b[2] + o.gyro[2]}}, vel_b);
    const V3 gb = mul_t(c_nb, V3{{0.0, 0.0, g}});
    for (int i = 0; i < 3; ++i) o.acc[i] = vel_dot_b[i] + cor[i] - gb[i];
    return o;
}

// pathgen.parse_motion_def (pathgen.py:413-439): the command of a segment -> target attitude and body velocity; types 3 / 5
// take the attitude, 3 / 4 the velocity relative to the state at the segment's start.  false: no such command type.
inline bool motion_target(const double* row, const V3& att, const V3& vel_b, V3& tgt_a, V3& tgt_v) {
    if (!(row[0] == 1 || row[0] == 2 || row[0] == 3 || row[0] == 4 || row[0] == 5)) return false;
    const bool rel_a = (row[0] == 3 || row[0] == 5), rel_v = (row[0] == 3 || row[0] == 4);
    for (int i = 0; i < 3; ++i) {
        tgt_a[i] = rel_a ? att[i] + row[1 + i] : row[1 + i];
        tgt_v[i] = rel_v ? vel_b[i] + row[4 + i] : row[4 + i];
    }
    return true;
}

}  // namespace
}  // namespace ginsim

using namespace ginsim;

// ---- the leaves of path_gen / FreeIntegration.run under their own entry points (host code; what a hosted plugin calls)
extern "C" int ginsim_calc_true_sensor_output(const double* pos_n, const double* vel_b, const double* att, const double* c_nb,
                                              const double* vel_dot_b, const double* att_dot, int32_t ref_frame, double g,
                                              double* acc, double* gyro, double* vel_dot_n, double* pos_dot_n) {
    if (!pos_n || !vel_b || !att || !c_nb || !vel_dot_b || !att_dot || !acc || !gyro || !vel_dot_n || !pos_dot_n) {
        set_error("calc_true_sensor_output: NULL argument");
        return GINSIM_ERR_ARG;
    }
    M3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[i][j] = c_nb[3 * i + j];
    const TrueSensor o = true_sensor_output(V3{{pos_n[0], pos_n[1], pos_n[2]}}, V3{{vel_b[0], vel_b[1], vel_b[2]}},
                                            V3{{att[0], att[1], att[2]}}, c, V3{{vel_dot_b[0], vel_dot_b[1], vel_dot_b[2]}},
                                            V3{{att_dot[0], att_dot[1], att_dot[2]}}, ref_frame == 0 ? 0 : 1, g, true);
    for (int i = 0; i < 3; ++i) { acc[i] = o.acc[i]; gyro[i] = o.gyro[i]; vel_dot_n[i] = o.vel_dot_n[i]; pos_dot_n[i] = o.pos_dot[i]; }
    return GINSIM_OK;
}

extern "C" int ginsim_parse_motion_def(const double* seg, const double* att, const double* vel, double* att_com, double* vel_com) {
    if (!seg || !att || !vel || !att_com || !vel_com) { set_error("parse_motion_def: NULL argument"); return GINSIM_ERR_ARG; }
    V3 a, v;
    if (!motion_target(seg, V3{{att[0], att[1], att[2]}}, V3{{vel[0], vel[1], vel[2]}}, a, v)) {
        set_error("parse_motion_def: unsupported motion type %g", seg[0]);
        return GINSIM_ERR_ARG;
    }
    for (int i = 0; i < 3; ++i) { att_com[i] = a[i]; vel_com[i] = v[i]; }
    return GINSIM_OK;
}

// attitude.euler_update_zyx (attitude.py:679-721): Euler-rate integration over dt, pitch fold at +-pi/2, ONE +-2 pi wrap of yaw
// and roll.  The per-step form of what Att::step (ins_math.hpp) does on the device with cached trigonometry.
extern "C" int ginsim_euler_update_zyx(const double* x, const double* w, double dt, double* y) {
    if (!x || !w || !y) { set_error("euler_update_zyx: NULL argument"); return GINSIM_ERR_ARG; }
    const double cr = std::cos(x[2]), sr = std::sin(x[2]);
    const double t = w[2] * cr + w[1] * sr;
    double yaw = x[0] + (t / std::cos(x[1])) * dt;
    double pit = x[1] + (w[1] * cr - w[2] * sr) * dt;
    double rol = x[2] + (w[0] + t * std::tan(x[1])) * dt;
    if (pit > 0.5 * kPi) { pit = kPi - pit; yaw = yaw + kPi; rol = rol + kPi; }
    else if (pit < -0.5 * kPi) { pit = -kPi - pit; yaw = yaw + kPi; rol = rol + kPi; }
    if (yaw > kPi) yaw = yaw - 2.0 * kPi; else if (yaw < -kPi) yaw = yaw + 2.0 * kPi;
    if (rol > kPi) rol = rol - 2.0 * kPi; else if (rol < -kPi) rol = rol + 2.0 * kPi;
    y[0] = yaw; y[1] = pit; y[2] = rol;
    return GINSIM_OK;
}

extern "C" int ginsim_pathgen_capacity(const ginsim_pathgen_params* p, const double* md, int64_t* cap) {
    if (!p || !md || !cap || p->n_seg < 1 || !(p->fs > 0)) { set_error("pathgen: bad arguments"); return GINSIM_ERR_ARG; }
    int64_t total = 0;
    for (int i = 0; i < p->n_seg; ++i) {
        const double dur = md[9 * i + 7];
        if (dur < 0) {      // pathgen.py:117-119
            set_error("Time duration of %d-th command has negative time duration: %g.", i, dur);
            return GINSIM_ERR_ARG;
        }
        total += (int64_t)std::ceil(dur * p->fs);
    }
    if (total <= 0) {       // pathgen.py:124-125
        set_error("Total time duration in the motion definition file must be above 0.");
        return GINSIM_ERR_ARG;
    }
    *cap = total;
    return GINSIM_OK;
}

extern "C" int ginsim_pathgen(const ginsim_pathgen_params* p, const double* md, int64_t cap, double* imu, double* nav,
                              double* gps, double* odo, double* mag, int64_t* n_out, int64_t* m_out) {
    int64_t need = 0;
    const int rc = ginsim_pathgen_capacity(p, md, &need);
    if (rc != GINSIM_OK) return rc;
    if (!imu || !nav || !n_out) { set_error("pathgen: imu/nav/n_out must not be NULL"); return GINSIM_ERR_ARG; }
    if (cap < need) { set_error("pathgen: capacity %lld < required %lld", (long long)cap, (long long)need); return GINSIM_ERR_RANGE; }
    const bool want_gps = p->enable_gps && gps;
    if (p->enable_gps && !(p->fs_gps > 0)) { set_error("pathgen: fs_gps must be > 0 when GPS is enabled"); return GINSIM_ERR_ARG; }

    const double fs = p->fs, dt = 1.0 / fs;
    const double alpha = 0.9, beta = 1 - alpha;                  // pathgen.py:101-103
    const double max_acc = p->mobility[0], max_dw = p->mobility[1], max_w = p->mobility[2];
    const double kp = 5.0, kd = 10.0;                             // pathgen.py:107-108
    const int64_t gps_every = p->enable_gps ? (int64_t)std::nearbyint(fs / p->fs_gps) : 0;   // pathgen.py:136
    if (p->enable_gps && gps_every < 1) { set_error("pathgen: fs_gps above fs"); return GINSIM_ERR_ARG; }

    V3 pos0{{p->ini_pva[0], p->ini_pva[1], p->ini_pva[2]}};
    V3 vel_b{{p->ini_pva[3], p->ini_pva[4], p->ini_pva[5]}};
    V3 att{{p->ini_pva[6], p->ini_pva[7], p->ini_pva[8]}};
    M3 c_nb = body_to_nav(att);
    V3 vel_n = mul(c_nb, vel_b);
    V3 dpos{{0, 0, 0}};
    const double g0 = earth(pos0[0], pos0[2]).g;                 // pathgen.py:162-163
    if (p->ref_frame == 1) {                                      // pathgen.py:173-174, geoparams.py:70-87
        const double sl = std::sin(pos0[0]), cl = std::cos(pos0[0]);
        const double r = kRe / std::sqrt(1.0 - kEsq * sl * sl);
        const double rho = (r + pos0[2]) * cl;
        pos0 = V3{{rho * std::cos(pos0[1]), rho * std::sin(pos0[1]), (r * (1.0 - kEsq) + pos0[2]) * sl}};
    }
    const bool want_mag = p->enable_mag && mag;
    V3 geo_mag{{p->geo_mag_n[0], p->geo_mag_n[1], p->geo_mag_n[2]}};
    if (want_mag && p->ref_frame == 1) {                          // pathgen.py:169-171: remove the declination
        geo_mag[0] = std::sqrt(geo_mag[0] * geo_mag[0] + geo_mag[1] * geo_mag[1]);
        geo_mag[1] = 0.0;
    }
    V3 att_dot{{0, 0, 0}}, vel_dot_b{{0, 0, 0}};
    int64_t k = 0, kg = 0;
    double odo_dist = 0.0;

    for (int seg = 0; seg < p->n_seg; ++seg) {
        const double* row = md + 9 * seg;
        const long typ = std::lround(row[0]);                    // pathgen.py:178 (Python round: half-to-even; types are integers)
        const double vis = row[8];
        V3 tgt_a, tgt_v;                                          // parse_motion_def, pathgen.py:413-439
        if (!motion_target(row, att, vel_b, tgt_a, tgt_v)) {
            set_error("pathgen: unsupported motion type %g in segment %d", row[0], seg);
            return GINSIM_ERR_ARG;
        }
        V3 filt_a = att, filt_v = vel_b;                          // pathgen.py:191-192
        const double stop = (double)k + std::nearbyint(row[7] * fs);     // pathgen.py:122, 194 (round half-to-even)
        bool done = false;
        while ((double)k < stop && !done) {
            if (typ == 1) {                                       // pathgen.py:199-200
                for (int i = 0; i < 3; ++i) {
                    att_dot[i] = alpha * att_dot[i] + beta * tgt_a[i];
                    vel_dot_b[i] = alpha * vel_dot_b[i] + beta * tgt_v[i];
                }
            } else {                                              // pathgen.py:203-223
                V3 da, dv;
                for (int i = 0; i < 3; ++i) {
                    filt_a[i] = alpha * filt_a[i] + beta * tgt_a[i];
                    filt_v[i] = alpha * filt_v[i] + beta * tgt_v[i];
                    vel_dot_b[i] = clamp((filt_v[i] - vel_b[i]) / dt, max_acc);
                    const double acc = clamp(kp * (tgt_a[i] - att[i]) + kd * (0 - att_dot[i]), max_dw);
                    att_dot[i] = clamp(att_dot[i] + acc * dt, max_w);
                    da[i] = att[i] - tgt_a[i];
                    dv[i] = vel_b[i] - tgt_v[i];
                }
                if (norm(da) < 1e-4 && norm(dv) < 1e-4) done = true;
            }
            // calc_true_sensor_output, pathgen.py:331-411
            const V3 pos{{pos0[0] + dpos[0], pos0[1] + dpos[1], pos0[2] + dpos[2]}};
            const TrueSensor ts = true_sensor_output(pos, vel_b, att, c_nb, vel_dot_b, att_dot, p->ref_frame, g0, false);
            const V3& gyro = ts.gyro;
            const V3& pos_dot = ts.pos_dot;
            // emit rows, pathgen.py:244-303
            double* qi = imu + 7 * k;
            double* qn = nav + 10 * k;
            const V3 eul = euler_range(att);
            qi[0] = (double)k;
            qn[0] = (double)k;
            for (int i = 0; i < 3; ++i) {
                qi[1 + i] = ts.acc[i];
                qi[4 + i] = gyro[i];
                qn[1 + i] = pos[i];
                qn[4 + i] = vel_n[i];
                qn[7 + i] = eul[i];
            }
            if (want_mag) {                                       // pathgen.py:273-279
                const V3 mb = mul_t(c_nb, geo_mag);
                double* qm = mag + 4 * k;
                qm[0] = (double)k; qm[1] = mb[0]; qm[2] = mb[1]; qm[3] = mb[2];
            }
            if (odo) {
                double* qo = odo + 5 * k;
                qo[0] = (double)k; qo[1] = odo_dist; qo[2] = vel_b[0]; qo[3] = vel_b[1]; qo[4] = vel_b[2];
            }
            if (want_gps && (k % gps_every) == 0) {
                double* qg = gps + 8 * kg;
                qg[0] = (double)k;
                for (int i = 0; i < 3; ++i) { qg[1 + i] = pos[i]; qg[4 + i] = vel_n[i]; }
                qg[7] = vis;
                ++kg;
            }
            // integrate, pathgen.py:306-311
            odo_dist = odo_dist + norm(vel_b) * dt;
            for (int i = 0; i < 3; ++i) {
                dpos[i] = dpos[i] + pos_dot[i] * dt;
                vel_b[i] = vel_b[i] + vel_dot_b[i] * dt;
                att[i] = att[i] + att_dot[i] * dt;
            }
            c_nb = body_to_nav(att);
            vel_n = mul(c_nb, vel_b);
            ++k;
        }
        if (done) { att_dot = V3{{0, 0, 0}}; vel_dot_b = V3{{0, 0, 0}}; }     // pathgen.py:317-319
    }
    *n_out = k;
    if (m_out) *m_out = kg;
    return GINSIM_OK;
}
