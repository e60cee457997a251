/*
 * ginsim_oracle.c -- plain-C CPU restatement of the reference hot path.  TEST ORACLE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (gnss-ins-sim_amd/) never does and has no CPU fallback.
 *
 * One Monte-Carlo run is computed exactly the way the reference computes it -- generate the whole
 * sensor series first, then integrate it sample by sample -- with scalar loops, so this file is also
 * the honest "CPU port" baseline (OpenMP over runs when built with -fopenmp):
 *
 *   bias_drift / acc_gen / gyro_gen   gnss_ins_sim/pathgen/pathgen.py:441-594
 *   odo_gen                           gnss_ins_sim/pathgen/pathgen.py:627-641
 *   FreeIntegration.run               demo_algorithms/free_integration.py:63-174
 *   FreeIntegration.run (odometer)    demo_algorithms/free_integration_odo.py:63-160
 *   euler_update_zyx / euler2dcm      gnss_ins_sim/attitude/attitude.py:679-721 / 344-371
 *   geo_param / lla2ecef              gnss_ins_sim/geoparams/geoparams.py:25-53 / 70-87
 *   array_error (end point)           gnss_ins_sim/sim/ins_data_manager.py:537-541, 737
 *
 * The noise source is the engine's Philox4x32-7 + single-precision inverse-CDF stream (see oracle/philox.py for the
 * definition, which is exact to the bit, and why the reference's own np.random stream is "parity unpinned").
 * Pinned against the NumPy oracle (itself pinned against the executed reference) in
 * tests/test_oracle_c.py.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.14159265358979323846
#define RE 6378137.0
#define FLAT (1.0 / 298.257223563)
#define ECC 0.0818191908426215
#define ESQ (ECC * ECC)
#define WIE 7292115e-11

/* ---------------------------------------------------------------- Philox4x32-7 + Box-Muller */
static void philox4x32_7(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 7; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* The coefficient table of the generator: the committed numbers the device uses (csrc/normal_tables.inc, made by
 * tools/gen_normal_tables.py; oracle/philox.py reads the same file). */
static const uint32_t normal_table_bits[31 * 8 * 4] = {
#include "normal_tables.inc"
};
static float bits_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* one standard normal from one 32-bit word: piecewise-cubic inversion of the upper tail probability t = m 2^-32,
 * defined operation by operation (oracle/philox.py normal_icdf) */
static float normal_icdf(uint32_t w) {
    const uint32_t m = (w & 0x7fffffffu) | 1u;
    const int lz = __builtin_clz(m);                                   /* 1 .. 31 */
    const uint32_t y = m << lz;
    const uint32_t* c = normal_table_bits + 4 * ((lz - 1) * 8 + (int)((y >> 28) & 7u));
    const float x = bits_f32(0x3f800000u | ((y << 4) >> 9));
    const float z = fmaf(fmaf(fmaf(bits_f32(c[3]), x, bits_f32(c[2])), x, bits_f32(c[1])), x, bits_f32(c[0]));
    return bits_f32((f32_bits(z) & 0x7fffffffu) | (w & 0x80000000u));
}

/* stream s at sample j = half (s & 1) of block (j, s >> 1): z0 from its first word, z1 from its second */
static void normal_pair_f32(uint64_t seed, uint64_t run, uint32_t stream, uint32_t j, float* z0, float* z1) {
    uint32_t W[4] = {j, stream >> 1, (uint32_t)run, (uint32_t)(run >> 32)};
    philox4x32_7(W, (uint32_t)seed, (uint32_t)(seed >> 32));
    *z0 = normal_icdf((stream & 1u) ? W[2] : W[0]);
    *z1 = normal_icdf((stream & 1u) ? W[3] : W[1]);
}

/* the fp64 path widens the single-precision normals (exactly) */
static void normal_pair(uint64_t seed, uint64_t run, uint32_t stream, uint32_t j, double* z0, double* z1) {
    float a0, a1;
    normal_pair_f32(seed, run, stream, j, &a0, &a1);
    *z0 = (double)a0;
    *z1 = (double)a1;
}

void oracle_normals(uint64_t seed, uint64_t run, uint32_t stream, int64_t count, double* z0, double* z1) {
    for (int64_t j = 0; j < count; ++j) normal_pair(seed, run, stream, (uint32_t)j, &z0[j], &z1[j]);
}

/* ---------------------------------------------------------------- leaves */
typedef struct { double rm, rn, g, sl, cl; } geo_t;

static geo_t geo_param(double lat, double h) {
    geo_t e;
    e.sl = sin(lat);
    e.cl = cos(lat);
    double s2 = e.sl * e.sl;
    e.rm = (RE * (1 - ESQ)) / (sqrt(1.0 - ESQ * s2) * (1.0 - ESQ * s2));
    e.rn = RE / sqrt(1.0 - ESQ * s2);
    double g1 = 9.7803253359 * (1 + 0.00193185265241 * s2) / sqrt(1.0 - ESQ * s2);
    e.g = g1 * (1.0 - (2.0 / RE) * (1.0 + FLAT + 0.00344978650684 - 2.0 * FLAT * s2) * h + 3.0 * h * h / RE / RE);
    return e;
}

static void lla2ecef(const double* lla, double* xyz) {
    double sl = sin(lla[0]), cl = cos(lla[0]);
    double r = RE / sqrt(1.0 - ESQ * sl * sl);
    double rho = (r + lla[2]) * cl;
    xyz[0] = rho * cos(lla[1]);
    xyz[1] = rho * sin(lla[1]);
    xyz[2] = (r * (1.0 - ESQ) + lla[2]) * sl;
}

static void euler2dcm(const double* a, double c[3][3]) {   /* n -> b */
    double cy = cos(a[0]), cp = cos(a[1]), cr = cos(a[2]);
    double sy = sin(a[0]), sp = sin(a[1]), sr = sin(a[2]);
    c[0][0] = cp * cy;                 c[0][1] = cp * sy;                 c[0][2] = -sp;
    c[1][0] = sr * sp * cy - cr * sy;  c[1][1] = sr * sp * sy + cr * cy;  c[1][2] = cp * sr;
    c[2][0] = sp * cr * cy + sy * sr;  c[2][1] = sp * cr * sy - cy * sr;  c[2][2] = cp * cr;
}

static void euler_update_zyx(const double* x, const double* w, double dt, double* y) {
    double c = cos(x[2]), s = sin(x[2]);
    double yaw_dot = (w[2] * c + w[1] * s) / cos(x[1]);
    double pit_dot = w[1] * c - w[2] * s;
    double rol_dot = w[0] + (w[2] * c + w[1] * s) * tan(x[1]);
    y[0] = x[0] + yaw_dot * dt;
    y[1] = x[1] + pit_dot * dt;
    y[2] = x[2] + rol_dot * dt;
    if (y[1] > 0.5 * PI) { y[1] = PI - y[1]; y[0] += PI; y[2] += PI; }
    else if (y[1] < -0.5 * PI) { y[1] = -PI - y[1]; y[0] += PI; y[2] += PI; }
    if (y[0] > PI) y[0] -= 2 * PI; else if (y[0] < -PI) y[0] += 2 * PI;
    if (y[2] > PI) y[2] -= 2 * PI; else if (y[2] < -PI) y[2] += 2 * PI;
}

static void mat_vec(double c[3][3], const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = c[i][0] * v[0] + c[i][1] * v[1] + c[i][2] * v[2];
}
static void mat_t_vec(double c[3][3], const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = c[0][i] * v[0] + c[1][i] * v[1] + c[2][i] * v[2];
}
static void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double angle_range_pi(double x) {
    x = fmod(x, 2 * PI);
    if (x < 0) x += 2 * PI;
    return x > PI ? x - 2 * PI : x;
}

/* ---------------------------------------------------------------- sensor errors */
typedef struct {
    double bias[3], gm_a[3], gm_b[3], white[3];
    int32_t white_drift[3];
    int32_t reserved;
} sensor_model_t;

/* meas[n][3] = ref + b + drift + white ; zd/zw stream ids are (s0,s0+1,s0+2) as in oracle/philox.py */
static void sensor_gen(uint64_t seed, uint64_t run, uint32_t s0, int64_t n, const double* ref,
                       const sensor_model_t* m, double* meas) {
    double d[3] = {0, 0, 0};
    for (int64_t j = 0; j < n; ++j) {
        double zd[3], zw[3];
        normal_pair(seed, run, s0, (uint32_t)j, &zd[0], &zd[1]);
        normal_pair(seed, run, s0 + 1, (uint32_t)j, &zd[2], &zw[0]);
        normal_pair(seed, run, s0 + 2, (uint32_t)j, &zw[1], &zw[2]);
        for (int i = 0; i < 3; ++i) {
            double dj = m->white_drift[i] ? m->gm_b[i] * zd[i] : d[i];
            meas[3 * j + i] = ref[3 * j + i] + m->bias[i] + dj + m->white[i] * zw[i];
            d[i] = m->gm_a[i] * d[i] + m->gm_b[i] * zd[i];
        }
    }
}

/* The vibration term of acc_gen / gyro_gen (pathgen.py:476-492 / 538-556), added last (:500, :562).  type 1 'random':
 * amp N[j] with the normals of streams s_xy, s_xy + 1 (oracle/philox.py vib_normals); type 2 'sinusoidal':
 * amp sin(omega_dt j + phase), phase = (u 2) pi with u = word 2^-32 of the phase block for gyro_gen (random_phase), 0 for acc_gen. */
typedef struct {
    int32_t type, random_phase;
    double amp[3];
    double omega_dt;
} vibration_t;

static void vibration_add(uint64_t seed, uint64_t run, uint32_t s_xy, uint32_t s_phase, int64_t n, const vibration_t* v,
                          double* meas) {
    if (!v || v->type == 0) return;
    if (v->type == 1) {
        for (int64_t j = 0; j < n; ++j) {
            double z[4];
            normal_pair(seed, run, s_xy, (uint32_t)j, &z[0], &z[1]);
            normal_pair(seed, run, s_xy + 1, (uint32_t)j, &z[2], &z[3]);
            for (int i = 0; i < 3; ++i) meas[3 * j + i] = meas[3 * j + i] + v->amp[i] * z[i];
        }
        return;
    }
    double phase[3] = {0, 0, 0};
    if (v->random_phase) {
        uint32_t W[4] = {0u, s_phase >> 1, (uint32_t)run, (uint32_t)(run >> 32)};
        philox4x32_7(W, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int i = 0; i < 3; ++i) phase[i] = ((double)W[i] * 0x1p-32 * 2) * PI;
    }
    for (int64_t j = 0; j < n; ++j) {
        const double cj = v->omega_dt * (double)j;
        for (int i = 0; i < 3; ++i) meas[3 * j + i] = meas[3 * j + i] + v->amp[i] * sin(v->random_phase ? cj + phase[i] : cj);
    }
}

/* ---------------------------------------------------------------- mechanisation of one run */
/* att/pos/vel are [n][3] scratch (or output) arrays; odo == NULL selects free_integration.py */
void oracle_free_integration(int ref_frame, double fs, int earth_rot, int64_t n, const double* gyro,
                             const double* accel, const double* odo, const double* ini, int has_g,
                             double* att, double* pos, double* vel) {
    double dt = 1.0 / fs, c[3][3], vb[3] = {ini[3], ini[4], ini[5]};
    att[0] = ini[6]; att[1] = ini[7]; att[2] = ini[8];
    euler2dcm(att, c);
    mat_t_vec(c, vb, vel);
    if (ref_frame == 1) {
        lla2ecef(ini, pos);
        double g = has_g ? ini[9] : geo_param(ini[0], ini[2]).g;
        double gn[3] = {0, 0, g};
        for (int64_t i = 1; i < n; ++i) {
            const double* w = gyro + 3 * (i - 1);
            euler_update_zyx(att + 3 * (i - 1), w, dt, att + 3 * i);
            if (odo) {
                vb[0] = odo[i - 1]; vb[1] = 0; vb[2] = 0;
            } else {
                double cg[3], wxv[3];
                mat_vec(c, gn, cg);
                cross3(w, vb, wxv);
                for (int k = 0; k < 3; ++k) vb[k] = vb[k] + (accel[3 * (i - 1) + k] + cg[k]) * dt - wxv[k] * dt;
            }
            euler2dcm(att + 3 * i, c);
            mat_t_vec(c, vb, vel + 3 * i);
            for (int k = 0; k < 3; ++k) pos[3 * i + k] = pos[3 * (i - 1) + k] + vel[3 * (i - 1) + k] * dt;
        }
    } else {
        pos[0] = ini[0]; pos[1] = ini[1]; pos[2] = ini[2];
        for (int64_t i = 1; i < n; ++i) {
            const double* p = pos + 3 * (i - 1);
            const double* v = vel + 3 * (i - 1);
            geo_t e = geo_param(p[0], p[2]);
            double rm_e = e.rm + p[2], rn_e = e.rn + p[2];
            double w_en[3] = {v[1] / rn_e, -v[0] / rm_e, -v[1] * e.sl / e.cl / rn_e};
            double w_ie[3] = {0, 0, 0};
            if (earth_rot) { w_ie[0] = WIE * e.cl; w_ie[2] = -WIE * e.sl; }
            double wsum[3] = {w_en[0] + w_ie[0], w_en[1] + w_ie[1], w_en[2] + w_ie[2]}, wb[3], w_nb[3];
            mat_vec(c, wsum, wb);
            for (int k = 0; k < 3; ++k) w_nb[k] = gyro[3 * (i - 1) + k] - wb[k];
            euler_update_zyx(att + 3 * (i - 1), w_nb, dt, att + 3 * i);
            if (!odo) {
                double an[3], cor[3], w2[3] = {2 * w_ie[0] + w_en[0], 2 * w_ie[1] + w_en[1], 2 * w_ie[2] + w_en[2]};
                double gn[3] = {0, 0, has_g ? ini[9] : e.g};
                mat_t_vec(c, accel + 3 * (i - 1), an);
                cross3(w2, v, cor);
                for (int k = 0; k < 3; ++k) vel[3 * i + k] = v[k] + (an[k] + gn[k] - cor[k]) * dt;
            }
            pos[3 * i + 0] = p[0] + v[0] / rm_e * dt;
            pos[3 * i + 1] = p[1] + v[1] / rn_e / e.cl * dt;
            pos[3 * i + 2] = p[2] + (-v[2]) * dt;
            euler2dcm(att + 3 * i, c);
            if (odo) {
                double vb2[3] = {odo[i - 1], 0, 0};
                mat_t_vec(c, vb2, vel + 3 * i);
            }
        }
    }
}

/* ---------------------------------------------------------------- Monte-Carlo batch */
typedef struct {
    int64_t n, runs;
    uint64_t run_offset, seed;
    double fs;
    int32_t ref_frame, algo_odo, earth_rot, n_ini;
    uint64_t ini_first;
    int32_t ini_has_g, reserved;
    sensor_model_t accel, gyro;
    double odo_scale, odo_stdv;
    double ref_end[9];
} oracle_mc_t;

/* end_err [runs][9]; traj (optional) [n_keep][n][9] for the first n_keep runs; sens (optional) [n_keep][n][6] */
int oracle_mc_run_vib(const oracle_mc_t* p, const vibration_t* vib_accel, const vibration_t* vib_gyro, const double* ini_table,
                      const double* ref_accel, const double* ref_gyro, const double* ref_odo, double* end_err, int64_t n_keep,
                      double* traj, double* sens);

int oracle_mc_run(const oracle_mc_t* p, const double* ini_table, const double* ref_accel, const double* ref_gyro,
                  const double* ref_odo, double* end_err, int64_t n_keep, double* traj, double* sens) {
    return oracle_mc_run_vib(p, NULL, NULL, ini_table, ref_accel, ref_gyro, ref_odo, end_err, n_keep, traj, sens);
}

/* the same with the vibration terms of Sim(env=...) (either may be NULL) */
int oracle_mc_run_vib(const oracle_mc_t* p, const vibration_t* vib_accel, const vibration_t* vib_gyro, const double* ini_table,
                      const double* ref_accel, const double* ref_gyro, const double* ref_odo, double* end_err, int64_t n_keep,
                      double* traj, double* sens) {
    const int64_t n = p->n;
    int fail = 0;
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * 3 * n);
        double* gyr = (double*)malloc(sizeof(double) * 3 * n);
        double* odo = (double*)malloc(sizeof(double) * n);
        double* att = (double*)malloc(sizeof(double) * 3 * n);
        double* pos = (double*)malloc(sizeof(double) * 3 * n);
        double* vel = (double*)malloc(sizeof(double) * 3 * n);
        if (!acc || !gyr || !odo || !att || !pos || !vel) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t r = 0; r < p->runs; ++r) {
                const uint64_t run = p->run_offset + (uint64_t)r;
                const uint64_t call = p->ini_first + (uint64_t)r;
                const double* ini = ini_table + 10 * (call < (uint64_t)p->n_ini ? call : 0);
                sensor_gen(p->seed, run, 0, n, ref_accel, &p->accel, acc);      /* ins_sim.py:491-493 */
                sensor_gen(p->seed, run, 3, n, ref_gyro, &p->gyro, gyr);        /* ins_sim.py:494-496 */
                vibration_add(p->seed, run, 10, 26, n, vib_accel, acc);
                vibration_add(p->seed, run, 12, 24, n, vib_gyro, gyr);
                if (p->algo_odo) {                                              /* ins_sim.py:504-506 */
                    for (int64_t j = 0; j < n; ++j) {
                        double z0, z1;
                        normal_pair(p->seed, run, 6, (uint32_t)j, &z0, &z1);
                        odo[j] = p->odo_scale * ref_odo[j] + p->odo_stdv * z0;
                    }
                }
                oracle_free_integration(p->ref_frame, p->fs, p->earth_rot, n, gyr, acc, p->algo_odo ? odo : NULL,
                                        ini, p->ini_has_g, att, pos, vel);
                double* e = end_err + 9 * r;
                for (int k = 0; k < 3; ++k) {
                    e[k] = angle_range_pi(att[3 * (n - 1) + k] - p->ref_end[k]);
                    e[3 + k] = pos[3 * (n - 1) + k] - p->ref_end[3 + k];
                    e[6 + k] = vel[3 * (n - 1) + k] - p->ref_end[6 + k];
                }
                if (r < n_keep) {
                    if (traj)
                        for (int64_t j = 0; j < n; ++j)
                            for (int k = 0; k < 3; ++k) {
                                traj[(r * n + j) * 9 + k] = att[3 * j + k];
                                traj[(r * n + j) * 9 + 3 + k] = pos[3 * j + k];
                                traj[(r * n + j) * 9 + 6 + k] = vel[3 * j + k];
                            }
                    if (sens)
                        for (int64_t j = 0; j < n; ++j)
                            for (int k = 0; k < 3; ++k) {
                                sens[(r * n + j) * 6 + k] = acc[3 * j + k];
                                sens[(r * n + j) * 6 + 3 + k] = gyr[3 * j + k];
                            }
                }
            }
        }
        free(acc); free(gyr); free(odo); free(att); free(pos); free(vel);
    }
    return fail ? -1 : 0;
}


/* ================================================================ fp32 path (BASELINE config 5)
 *
 * Float restatement of the SAME reference functions (acc_gen / gyro_gen / bias_drift, odo_gen, FreeIntegration.run for both
 * plugins and both frames), written to the numerical scheme of the fp32 HIP kernel (csrc/mc_kernel_f32.hip) OPERATION BY
 * OPERATION: every product, sum, quotient and square root is one IEEE single-precision operation in the kernel's order
 * (this file is compiled with -ffp-contract=off; fused multiply-adds are the explicit fmaf / fma calls), the attitude
 * carries Kahan-compensated Euler angles and a cached sin/cos that is rotated by the step and re-evaluated exactly
 * (sincos_def: fp64 quadrant reduction + Taylor polynomials, rounded to float once) every 32 steps, on a pitch fold and
 * for steps > 0.25 rad; position is accumulated in fp64 (ref_frame 1: the displacement from the initial ECEF position).
 * The kernel's sensor series and trajectories must equal this function's BIT FOR BIT (tests/test_gpu_fp32.py); against
 * the reference's fp64 outputs it is held to the stated fp32 tolerances (tests/test_oracle_c.py, reference-executed
 * goldens T1 / T2 / T3).  The noise is the fp64 path's: the same single-precision normals, not widened. */
#define TRIG_RESYNC 32
static const float PI_F = 3.14159265358979323846f, HALF_PI_F = 1.57079632679489661923f;
static const float TWO_PI_HI = 6.28318548202514648438f, TWO_PI_LO = -1.74845553146951715e-07f;

static void sincos_def(double x, float* sn, float* cs) {
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double t = r * r;
    double ps = -1.0 / 39916800.0;
    ps = fma(ps, t, 1.0 / 362880.0);
    ps = fma(ps, t, -1.0 / 5040.0);
    ps = fma(ps, t, 1.0 / 120.0);
    ps = fma(ps, t, -1.0 / 6.0);
    const double s = fma(r * t, ps, r);
    double pc = 1.0 / 479001600.0;
    pc = fma(pc, t, -1.0 / 3628800.0);
    pc = fma(pc, t, 1.0 / 40320.0);
    pc = fma(pc, t, -1.0 / 720.0);
    pc = fma(pc, t, 1.0 / 24.0);
    pc = fma(pc, t, -0.5);
    const double c = fma(t, pc, 1.0);
    const int q = (int)k & 3;
    const double so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    *sn = (float)((q & 2) ? -so : so);
    *cs = (float)(((q + 1) & 2) ? -co : co);
}

typedef struct { float v, comp; } kahan_t;
static void kadd(kahan_t* a, float inc) {
    const float t = inc - a->comp;
    const float s = a->v + t;
    a->comp = (s - a->v) - t;
    a->v = s;
}

static void rotate_f32(float d, float* s, float* c) {      /* sin/cos(a + d) from sin/cos(a), |d| <= 0.25 */
    const float t = d * d;
    const float sd = d * fmaf(t, fmaf(t, 8.3333333e-3f, -1.6666667e-1f), 1.0f);
    const float cm1 = t * fmaf(t, fmaf(t, -1.3888889e-3f, 4.1666667e-2f), -0.5f);
    const float s0 = *s, c0 = *c;
    *s = fmaf(c0, sd, fmaf(s0, cm1, s0));
    *c = fmaf(-s0, sd, fmaf(c0, cm1, c0));
}

typedef struct { kahan_t yaw, pit, rol; float sy, cy, sp, cp, sr, cr; } att32_t;

static void att_resync(att32_t* a) {
    sincos_def((double)a->yaw.v, &a->sy, &a->cy);
    sincos_def((double)a->pit.v, &a->sp, &a->cp);
    sincos_def((double)a->rol.v, &a->sr, &a->cr);
}

/* attitude.euler2dcm(., 'zyx') (attitude.py:360-368, n -> b) from the cached trig */
static void dcm_f32(const att32_t* a, float c[3][3]) {
    const float srsp = a->sr * a->sp, spcr = a->sp * a->cr;
    c[0][0] = a->cp * a->cy;                          c[0][1] = a->cp * a->sy;                          c[0][2] = -a->sp;
    c[1][0] = fmaf(srsp, a->cy, -(a->cr * a->sy));    c[1][1] = fmaf(srsp, a->sy, a->cr * a->cy);       c[1][2] = a->cp * a->sr;
    c[2][0] = fmaf(spcr, a->cy, a->sy * a->sr);       c[2][1] = fmaf(spcr, a->sy, -(a->cy * a->sr));    c[2][2] = a->cp * a->cr;
}
static void mat_vec_f32(float c[3][3], const float* v, float* o) {
    for (int i = 0; i < 3; ++i) o[i] = fmaf(c[i][2], v[2], fmaf(c[i][1], v[1], c[i][0] * v[0]));
}
static void mat_t_vec_f32(float c[3][3], const float* v, float* o) {
    for (int i = 0; i < 3; ++i) o[i] = fmaf(c[2][i], v[2], fmaf(c[1][i], v[1], c[0][i] * v[0]));
}
static void cross_f32(const float* a, const float* b, float* o) {
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

/* attitude.euler_update_zyx (attitude.py:679-721) */
static void att_step_f32(att32_t* a, const float* w, float dt, int resync) {
    const float q = fmaf(w[2], a->cr, w[1] * a->sr);
    const float icp = 1.0f / a->cp;
    const float dy = (q * icp) * dt;
    const float dp = fmaf(w[1], a->cr, -(w[2] * a->sr)) * dt;
    const float dr = fmaf(q, a->sp * icp, w[0]) * dt;
    kadd(&a->yaw, dy); kadd(&a->pit, dp); kadd(&a->rol, dr);
    const int fold = (a->pit.v > HALF_PI_F) || (a->pit.v < -HALF_PI_F);
    if (fold) {
        a->pit.v = a->pit.v > 0.f ? PI_F - a->pit.v : -PI_F - a->pit.v;
        a->pit.comp = 0.f;
        kadd(&a->yaw, PI_F); kadd(&a->rol, PI_F);
    }
    if (a->yaw.v > PI_F) { kadd(&a->yaw, -TWO_PI_HI); kadd(&a->yaw, -TWO_PI_LO); }
    else if (a->yaw.v < -PI_F) { kadd(&a->yaw, TWO_PI_HI); kadd(&a->yaw, TWO_PI_LO); }
    if (a->rol.v > PI_F) { kadd(&a->rol, -TWO_PI_HI); kadd(&a->rol, -TWO_PI_LO); }
    else if (a->rol.v < -PI_F) { kadd(&a->rol, TWO_PI_HI); kadd(&a->rol, TWO_PI_LO); }
    const float big = fmaxf(fabsf(dy), fmaxf(fabsf(dp), fabsf(dr)));
    if (resync || fold || !(big <= 0.25f)) {
        att_resync(a);
    } else {
        rotate_f32(dy, &a->sy, &a->cy); rotate_f32(dp, &a->sp, &a->cp); rotate_f32(dr, &a->sr, &a->cr);
    }
}

/* float sensor series of one run: meas[n][3] (pathgen.py:441-594), the normals of streams s0 .. s0+2 as they are */
static void sensor_gen_f32(uint64_t seed, uint64_t run, uint32_t s0, int64_t n, const double* ref,
                           const sensor_model_t* m, float* meas) {
    float d[3] = {0.f, 0.f, 0.f};
    for (int64_t j = 0; j < n; ++j) {
        float zd[3], zw[3];
        normal_pair_f32(seed, run, s0, (uint32_t)j, &zd[0], &zd[1]);
        normal_pair_f32(seed, run, s0 + 1, (uint32_t)j, &zd[2], &zw[0]);
        normal_pair_f32(seed, run, s0 + 2, (uint32_t)j, &zw[1], &zw[2]);
        for (int i = 0; i < 3; ++i) {
            const float bz = (float)m->gm_b[i] * zd[i];
            const float dj = m->white_drift[i] ? bz : d[i];
            const float base = ((float)ref[3 * j + i] + (float)m->bias[i]) + dj;
            meas[3 * j + i] = fmaf((float)m->white[i], zw[i], base);
            d[i] = fmaf((float)m->gm_a[i], d[i], bz);
        }
    }
}

/* FreeIntegration.run in float (free_integration.py:63-174; odo != NULL: free_integration_odo.py:63-160).
 * att / vel [n][3] float; dpos [n][3] float = position MINUS the initial position (ECEF displacement for ref_frame 1,
 * lat / lon / alt differences for ref_frame 0); end_pos[3] = the fp64 position at the last sample. */
void oracle_free_integration_f32(int ref_frame, double fs, int earth_rot, int64_t n, const float* gyro,
                                 const float* accel, const float* odo, const double* ini, int has_g,
                                 float* att, float* dpos, float* vel, double* end_pos) {
    const float dt = (float)(1.0 / fs);
    att32_t a;
    a.yaw.v = (float)ini[6]; a.pit.v = (float)ini[7]; a.rol.v = (float)ini[8];
    a.yaw.comp = a.pit.comp = a.rol.comp = 0.f;
    att_resync(&a);
    float c[3][3];
    kahan_t vb[3], vn[3];
    float vb0[3] = {(float)ini[3], (float)ini[4], (float)ini[5]}, v[3];
    for (int k = 0; k < 3; ++k) { vb[k].v = vb0[k]; vb[k].comp = 0.f; }
    dcm_f32(&a, c);
    mat_t_vec_f32(c, vb0, v);
    for (int k = 0; k < 3; ++k) { vn[k].v = v[k]; vn[k].comp = 0.f; }
    double pos[3], pos0[3];
    float g, sl, cl;
    if (ref_frame == 1) {
        lla2ecef(ini, pos0);
        pos[0] = pos[1] = pos[2] = 0.0;
        g = (float)(has_g ? ini[9] : geo_param(ini[0], ini[2]).g);
    } else {
        for (int k = 0; k < 3; ++k) pos0[k] = pos[k] = ini[k];
        g = has_g ? (float)ini[9] : 0.f;
    }
    sincos_def(ini[0], &sl, &cl);
    for (int64_t i = 0;; ++i) {
        att[3 * i] = a.yaw.v; att[3 * i + 1] = a.pit.v; att[3 * i + 2] = a.rol.v;
        for (int k = 0; k < 3; ++k) {
            dpos[3 * i + k] = ref_frame == 1 ? (float)pos[k] : (float)(pos[k] - pos0[k]);
            vel[3 * i + k] = v[k];
        }
        if (i == n - 1) break;
        const float* w = gyro + 3 * i;
        const int resync = ((i + 1) & (TRIG_RESYNC - 1)) == 0;
        if (ref_frame == 1) {
            const float v_prev[3] = {v[0], v[1], v[2]};
            if (!odo) {
                const float gb[3] = {-a.sp, a.cp * a.sr, a.cp * a.cr};          /* C . [0, 0, 1] */
                const float vbv[3] = {vb[0].v, vb[1].v, vb[2].v};
                float wxv[3];
                cross_f32(w, vbv, wxv);
                for (int k = 0; k < 3; ++k) kadd(&vb[k], (fmaf(gb[k], g, accel[3 * i + k]) - wxv[k]) * dt);
            }
            att_step_f32(&a, w, dt, resync);
            if (odo) {
                v[0] = (a.cp * a.cy) * odo[i]; v[1] = (a.cp * a.sy) * odo[i]; v[2] = (-a.sp) * odo[i];
            } else {
                const float vbv[3] = {vb[0].v, vb[1].v, vb[2].v};
                dcm_f32(&a, c);
                mat_t_vec_f32(c, vbv, v);
            }
            for (int k = 0; k < 3; ++k) pos[k] += (double)(v_prev[k] * dt);
        } else {
            /* geoparams.geo_param (geoparams.py:25-53) in float on the cached sin / cos of the latitude */
            const float h = (float)pos[2];
            const float s2 = sl * sl;
            const float qq = fmaf(-(float)ESQ, s2, 1.0f);
            const float sq = sqrtf(qq);
            const float rn = (float)RE / sq;
            const float rm = (float)(RE * (1.0 - ESQ)) / (qq * sq);
            const float g1 = ((float)9.7803253359 * fmaf((float)0.00193185265241, s2, 1.0f)) / sq;
            const float gh = fmaf((float)(3.0 / (RE * RE)), h * h,
                                  fmaf(-((float)(2.0 / RE) * fmaf(-2.0f * (float)FLAT, s2, (float)(1.0 + FLAT + 0.00344978650684))), h, 1.0f));
            const float gm = g1 * gh;
            const float irm = 1.0f / (rm + h), irn = 1.0f / (rn + h), icl = 1.0f / cl;
            const float w_en[3] = {v[1] * irn, -(v[0] * irm), -(((v[1] * sl) * icl) * irn)};
            float w_ie[3] = {0.f, 0.f, 0.f};
            if (earth_rot) { w_ie[0] = (float)WIE * cl; w_ie[2] = -((float)WIE * sl); }
            const float wsum[3] = {w_en[0] + w_ie[0], w_en[1] + w_ie[1], w_en[2] + w_ie[2]};
            float wb[3], w_nb[3];
            dcm_f32(&a, c);
            mat_vec_f32(c, wsum, wb);
            for (int k = 0; k < 3; ++k) w_nb[k] = w[k] - wb[k];
            if (!odo) {
                float an[3], cor[3];
                const float w2[3] = {fmaf(2.f, w_ie[0], w_en[0]), fmaf(2.f, w_ie[1], w_en[1]), fmaf(2.f, w_ie[2], w_en[2])};
                mat_t_vec_f32(c, accel + 3 * i, an);
                cross_f32(w2, v, cor);
                const float gg = has_g ? g : gm;
                kadd(&vn[0], (an[0] - cor[0]) * dt);
                kadd(&vn[1], (an[1] - cor[1]) * dt);
                kadd(&vn[2], ((an[2] + gg) - cor[2]) * dt);
            }
            att_step_f32(&a, w_nb, dt, resync);
            const float dlat = (v[0] * irm) * dt;
            pos[0] += (double)dlat;
            pos[1] += (double)(((v[1] * irn) * icl) * dt);
            pos[2] += (double)(-(v[2] * dt));
            if (resync) sincos_def(pos[0], &sl, &cl);
            else rotate_f32(dlat, &sl, &cl);
            if (odo) {
                v[0] = (a.cp * a.cy) * odo[i]; v[1] = (a.cp * a.sy) * odo[i]; v[2] = (-a.sp) * odo[i];
            } else {
                v[0] = vn[0].v; v[1] = vn[1].v; v[2] = vn[2].v;
            }
        }
    }
    for (int k = 0; k < 3; ++k) end_pos[k] = ref_frame == 1 ? pos0[k] + pos[k] : pos[k];
}

/* the given-data boundary in float: fp64 series rounded to float as they are read (the kernel's GIVEN mode) */
void oracle_free_integration_f32_given(int ref_frame, double fs, int earth_rot, int64_t n, const double* gyro,
                                       const double* accel, const double* odo, const double* ini, int has_g,
                                       float* att, float* dpos, float* vel, double* end_pos) {
    float* g = (float*)malloc(sizeof(float) * 3 * n);
    float* a = (float*)malloc(sizeof(float) * 3 * n);
    float* o = (float*)malloc(sizeof(float) * n);
    for (int64_t i = 0; i < 3 * n; ++i) { g[i] = (float)gyro[i]; a[i] = accel ? (float)accel[i] : 0.f; }
    for (int64_t i = 0; i < n; ++i) o[i] = odo ? (float)odo[i] : 0.f;
    oracle_free_integration_f32(ref_frame, fs, earth_rot, n, g, a, odo ? o : NULL, ini, has_g, att, dpos, vel, end_pos);
    free(g); free(a); free(o);
}

/* end_err [runs][9] (double); traj (optional) [n_keep][n][9] float = att3, position displacement3, vel3; sens (optional)
 * [n_keep][n][6] float; odo_out (optional) [n_keep][n] float */
/* the vibration term in single precision, as csrc/mc_kernel_f32.hip add_vibration defines it: amplitudes rounded to float;
 * 'random' o = fma(amp, z, o); 'sinusoidal' the angle omega_dt * j (+ phase) in fp64, its sine by sincos_def, o = fma(amp, sin, o) */
static void vibration_add_f32(uint64_t seed, uint64_t run, uint32_t s_xy, uint32_t s_phase, int64_t n, const vibration_t* v,
                              float* meas) {
    if (!v || v->type == 0) return;
    const float amp[3] = {(float)v->amp[0], (float)v->amp[1], (float)v->amp[2]};
    if (v->type == 1) {
        for (int64_t j = 0; j < n; ++j) {
            float z[4];
            normal_pair_f32(seed, run, s_xy, (uint32_t)j, &z[0], &z[1]);
            normal_pair_f32(seed, run, s_xy + 1, (uint32_t)j, &z[2], &z[3]);
            for (int i = 0; i < 3; ++i) meas[3 * j + i] = fmaf(amp[i], z[i], meas[3 * j + i]);
        }
        return;
    }
    double phase[3] = {0, 0, 0};
    if (v->random_phase) {
        uint32_t W[4] = {0u, s_phase >> 1, (uint32_t)run, (uint32_t)(run >> 32)};
        philox4x32_7(W, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int i = 0; i < 3; ++i) phase[i] = ((double)W[i] * 0x1p-32 * 2) * PI;
    }
    for (int64_t j = 0; j < n; ++j) {
        const double cj = v->omega_dt * (double)j;
        for (int i = 0; i < 3; ++i) {
            float sn, cs;
            sincos_def(v->random_phase ? cj + phase[i] : cj, &sn, &cs);
            meas[3 * j + i] = fmaf(amp[i], sn, meas[3 * j + i]);
        }
    }
}

int oracle_mc_run_f32_vib(const oracle_mc_t* p, const vibration_t* vib_accel, const vibration_t* vib_gyro, const double* ini_table,
                          const double* ref_accel, const double* ref_gyro, const double* ref_odo, double* end_err, int64_t n_keep,
                          float* traj, float* sens, float* odo_out);

int oracle_mc_run_f32(const oracle_mc_t* p, const double* ini_table, const double* ref_accel, const double* ref_gyro,
                      const double* ref_odo, double* end_err, int64_t n_keep, float* traj, float* sens, float* odo_out) {
    return oracle_mc_run_f32_vib(p, NULL, NULL, ini_table, ref_accel, ref_gyro, ref_odo, end_err, n_keep, traj, sens, odo_out);
}

int oracle_mc_run_f32_vib(const oracle_mc_t* p, const vibration_t* vib_accel, const vibration_t* vib_gyro, const double* ini_table,
                          const double* ref_accel, const double* ref_gyro, const double* ref_odo, double* end_err, int64_t n_keep,
                          float* traj, float* sens, float* odo_out) {
    const int64_t n = p->n;
    int fail = 0;
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * 3 * n);
        float* gyr = (float*)malloc(sizeof(float) * 3 * n);
        float* odo = (float*)malloc(sizeof(float) * n);
        float* att = (float*)malloc(sizeof(float) * 3 * n);
        float* pos = (float*)malloc(sizeof(float) * 3 * n);
        float* vel = (float*)malloc(sizeof(float) * 3 * n);
        if (!acc || !gyr || !odo || !att || !pos || !vel) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t r = 0; r < p->runs; ++r) {
                const uint64_t run = p->run_offset + (uint64_t)r;
                const uint64_t call = p->ini_first + (uint64_t)r;
                const double* ini = ini_table + 10 * (call < (uint64_t)p->n_ini ? call : 0);
                sensor_gen_f32(p->seed, run, 0, n, ref_accel, &p->accel, acc);
                sensor_gen_f32(p->seed, run, 3, n, ref_gyro, &p->gyro, gyr);
                vibration_add_f32(p->seed, run, 10, 26, n, vib_accel, acc);
                vibration_add_f32(p->seed, run, 12, 24, n, vib_gyro, gyr);
                if (ref_odo) {
                    for (int64_t j = 0; j < n; ++j) {
                        float z0, z1;
                        normal_pair_f32(p->seed, run, 6, (uint32_t)j, &z0, &z1);
                        odo[j] = fmaf((float)p->odo_stdv, z0, (float)p->odo_scale * (float)ref_odo[j]);
                    }
                }
                double end_pos[3];
                oracle_free_integration_f32(p->ref_frame, p->fs, p->earth_rot, n, gyr, acc, p->algo_odo ? odo : NULL,
                                            ini, p->ini_has_g, att, pos, vel, end_pos);
                double* e = end_err + 9 * r;
                for (int k = 0; k < 3; ++k) {
                    e[k] = angle_range_pi((double)att[3 * (n - 1) + k] - p->ref_end[k]);
                    e[3 + k] = end_pos[k] - p->ref_end[3 + k];
                    e[6 + k] = (double)vel[3 * (n - 1) + k] - p->ref_end[6 + k];
                }
                if (r < n_keep) {
                    if (traj)
                        for (int64_t j = 0; j < n; ++j)
                            for (int k = 0; k < 3; ++k) {
                                traj[(r * n + j) * 9 + k] = att[3 * j + k];
                                traj[(r * n + j) * 9 + 3 + k] = pos[3 * j + k];
                                traj[(r * n + j) * 9 + 6 + k] = vel[3 * j + k];
                            }
                    if (sens)
                        for (int64_t j = 0; j < n; ++j)
                            for (int k = 0; k < 3; ++k) {
                                sens[(r * n + j) * 6 + k] = acc[3 * j + k];
                                sens[(r * n + j) * 6 + 3 + k] = gyr[3 * j + k];
                            }
                    if (odo_out && ref_odo)
                        for (int64_t j = 0; j < n; ++j) odo_out[r * n + j] = odo[j];
                }
            }
        }
        free(acc); free(gyr); free(odo); free(att); free(pos); free(vel);
    }
    return fail ? -1 : 0;
}

/* ---------------------------------------------------------------- Allan variance (allan.py:18-59) */
/* returns ntau; avar/tau sized >= 64 */
int oracle_allan_var(const double* x, int64_t n, double fs, double* avar, double* tau) {
    double ts = 1.0 / fs;
    int64_t mmax = (int64_t)floor(n / 9.0);
    if (mmax * ts < 1) return 0;
    int64_t mult[128];
    int nt = 0;
    int decades = (int)ceil(log10((double)mmax));
    double scale = 0.1;
    for (int i = 0; i < decades; ++i) {
        scale *= 10;
        for (int j = 1; j < 10; ++j) {
            int64_t m = (int64_t)(j * scale);
            if (m > mmax) break;
            mult[nt++] = m;
        }
    }
    for (int i = 0; i < nt; ++i) { avar[i] = 0; tau[i] = 0; }
    for (int i = 0; i < nt; ++i) {
        int64_t m = mult[i], nb = n / m;
        if (nb < 9) break;
        double prev = 0, acc = 0;
        for (int64_t b = 0; b < nb; ++b) {
            double s = 0;
            for (int64_t k = 0; k < m; ++k) s += x[b * m + k];
            s /= (double)m;
            if (b > 0) acc += (s - prev) * (s - prev);
            prev = s;
        }
        avar[i] = 0.5 / (double)(nb - 1) * acc;
        tau[i] = (double)m * ts;
    }
    return nt;
}
