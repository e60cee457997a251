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
 * The noise source is the engine's Philox4x32-7 + single-precision Box-Muller stream (see oracle/philox.py for the
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

/* The two fp32 tables of the generator: the committed numbers the device uses (csrc/normal_tables.inc, made by
 * tools/gen_normal_tables.py; oracle/philox.py builds them independently and the tests compare the bits). */
static const uint32_t normal_table_bits[256 * 4 + 512 * 2] = {
#include "normal_tables.inc"
};
static float bits_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* stream s at sample j = half (s & 1) of block (j, s >> 1); the transform is defined operation by operation in IEEE
 * single precision (no fused multiply-adds: this file is compiled with -ffp-contract=off) -- oracle/philox.py */
static void normal_pair(uint64_t seed, uint64_t run, uint32_t stream, uint32_t j, double* z0, double* z1) {
    uint32_t W[4] = {j, stream >> 1, (uint32_t)run, (uint32_t)(run >> 32)};
    philox4x32_7(W, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t a = (stream & 1u) ? W[2] : W[0], b = (stream & 1u) ? W[3] : W[1];
    /* radius: x = -2 ln u, u = (f32(a) + 1/2) 2^-32 */
    const float t = (float)a;
    const float u = (t + 0.5f) * 0x1.0p-32f;
    const uint32_t hx = f32_bits(u) + (0x3f800000u - 0x3f3504f3u);
    const float ef = (float)((int32_t)(hx >> 23) - 127);
    const uint32_t* lg = normal_table_bits + 4 * ((hx >> 15) & 255u);
    const float m = bits_f32((hx & 0x007fffffu) + 0x3f3504f3u);
    const float d = m - bits_f32(lg[0]);
    const float r = d * bits_f32(lg[1]);
    float q = r * (1.0f / 12.0f);
    q = q + 0.25f;
    const float r2 = r * r;
    q = q * r2;
    const float small = r + q;
    float x = ef * -1.3862943611198906f;
    x = x + bits_f32(lg[2]);
    x = x + small;
    const float rad = sqrtf(x);
    /* direction: sin, cos of 2 pi ((b & 0xffffff) + 1/2) 2^-24 */
    const uint32_t* sc = normal_table_bits + 256 * 4 + 2 * ((b >> 15) & 511u);
    const float sn_i = bits_f32(sc[0]), cs_i = bits_f32(sc[1]);
    float bb = (float)(int32_t)(b & 0x7fffu) + (0.5f - 16384.0f);
    bb = bb * 3.7450703562e-07f;               /* f32(2 pi 2^-24) */
    const float tt = bb * bb;
    float u1 = tt * (-1.0f / 6.0f);
    u1 = u1 * bb;
    const float sb = bb + u1;
    const float cm = tt * -0.5f;
    float p1 = cs_i * sb;
    const float p2 = sn_i * cm;
    p1 = p1 + p2;
    const float sn = sn_i + p1;
    float q1 = cs_i * cm;
    const float q2 = sn_i * sb;
    q1 = q1 - q2;
    const float cs = cs_i + q1;
    const float a0 = rad * cs, a1 = rad * sn;
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
int oracle_mc_run(const oracle_mc_t* p, const double* ini_table, const double* ref_accel, const double* ref_gyro,
                  const double* ref_odo, double* end_err, int64_t n_keep, double* traj, double* sens) {
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
