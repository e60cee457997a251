// fp32 variant of the fused Monte-Carlo kernel (BASELINE config 5).  Same algorithm, same launch geometry and
// the same SoA [component][sample][run] outputs as mc_kernel.hip, in single precision:
//
//   * attitude, body/NED velocity and all sensor arithmetic in fp32; the running sums (three Euler angles, three
//     velocity components) are Kahan-compensated so that 1e5 forward-Euler steps do not random-walk in the last bit;
//   * POSITION is accumulated in fp64 (3 adds per step): ECEF (4.7e6 m) and LLA radians do not fit fp32 (ulp 0.5 m /
//     6e-8 rad), and the series written to HBM is the fp32 DISPLACEMENT from the initial position;
//   * noise: the same Philox4x32-7 generator, but a Box-Muller pair takes a 23-bit radius uniform
//     u = (2 (w >> 9) + 1) 2^-24 and an 18-bit angle, so TWO blocks give the six pairs (12 normals) of a step, evaluated with the
//     hardware v_log_f32 / v_sin_f32 / v_cos_f32 units.  |z| <= 5.77 sigma.  This is a different (coarser) noise
//     stream than the fp64 path; run-by-run comparison with fp64 is therefore done noise-free and in given-data
//     form, and with noise the comparison is statistical (tests/test_gpu_fp32.py states the tolerances).
//
// Restates the same reference functions as mc_kernel.hip (pathgen.py:441-594, free_integration.py:63-174,
// free_integration_odo.py:63-160, ins_data_manager.py:537-541).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ginsim.h"
#include "ins_math.hpp"
#include "philox.hpp"

namespace ginsim {

namespace f32 {

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kTwoPiHi = 6.28318548202514648438f;        // float(2 pi)
constexpr float kTwoPiLo = -1.74845553146951715e-07f;      // 2 pi - float(2 pi)
constexpr float kHalfPiF = 1.57079632679489661923f;

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// compensated accumulator: value += inc with the rounding error carried in comp
struct Acc {
    float v, comp;
    __device__ __forceinline__ void add(float inc) {
        const float t = inc - comp;
        const float s = v + t;
        comp = (s - v) - t;
        v = s;
    }
};

__device__ __forceinline__ void rotate(float d, float& s, float& c) {      // sin/cos(a+d) from sin/cos(a), |d| <= 0.25
    const float t = d * d;
    const float sd = d * fmaf(t, fmaf(t, 8.3333333e-3f, -1.6666667e-1f), 1.0f);
    const float cm1 = t * fmaf(t, fmaf(t, -1.3888889e-3f, 4.1666667e-2f), -0.5f);
    const float s0 = s, c0 = c;
    s = fmaf(c0, sd, fmaf(s0, cm1, s0));
    c = fmaf(-s0, sd, fmaf(c0, cm1, c0));
}

struct Att {
    Acc yaw, pit, rol;
    float sy, cy, sp, cp, sr, cr;
    __device__ __forceinline__ void set(float y, float p, float r) {
        yaw = Acc{y, 0.f}; pit = Acc{p, 0.f}; rol = Acc{r, 0.f};
        sincosf(y, &sy, &cy); sincosf(p, &sp, &cp); sincosf(r, &sr, &cr);
    }
    __device__ __forceinline__ void resync() {
        sincosf(yaw.v, &sy, &cy); sincosf(pit.v, &sp, &cp); sincosf(rol.v, &sr, &cr);
    }
    __device__ __forceinline__ V3 to_body(const V3& v) const {
        return V3{cp * cy * v.x + cp * sy * v.y - sp * v.z,
                  (sr * sp * cy - cr * sy) * v.x + (sr * sp * sy + cr * cy) * v.y + cp * sr * v.z,
                  (sp * cr * cy + sy * sr) * v.x + (sp * cr * sy - cy * sr) * v.y + cp * cr * v.z};
    }
    __device__ __forceinline__ V3 to_nav(const V3& v) const {
        return V3{cp * cy * v.x + (sr * sp * cy - cr * sy) * v.y + (sp * cr * cy + sy * sr) * v.z,
                  cp * sy * v.x + (sr * sp * sy + cr * cy) * v.y + (sp * cr * sy - cy * sr) * v.z,
                  -sp * v.x + cp * sr * v.y + cp * cr * v.z};
    }
    __device__ __forceinline__ V3 down_in_body() const { return V3{-sp, cp * sr, cp * cr}; }
    __device__ __forceinline__ V3 fwd_in_nav() const { return V3{cp * cy, cp * sy, -sp}; }

    // attitude.euler_update_zyx (attitude.py:679-721) in fp32
    __device__ __forceinline__ void step(const V3& w, float dt, bool do_resync) {
        const float q = w.z * cr + w.y * sr;
        const float icp = __builtin_amdgcn_rcpf(cp);
        const float dy = q * icp * dt;
        const float dp = (w.y * cr - w.z * sr) * dt;
        const float dr = (w.x + q * (sp * icp)) * dt;
        yaw.add(dy); pit.add(dp); rol.add(dr);
        const bool fold = (pit.v > kHalfPiF) || (pit.v < -kHalfPiF);
        if (fold) {
            pit = Acc{pit.v > 0.f ? kPiF - pit.v : -kPiF - pit.v, 0.f};
            yaw.add(kPiF); rol.add(kPiF);
        }
        if (yaw.v > kPiF) { yaw.add(-kTwoPiHi); yaw.add(-kTwoPiLo); } else if (yaw.v < -kPiF) { yaw.add(kTwoPiHi); yaw.add(kTwoPiLo); }
        if (rol.v > kPiF) { rol.add(-kTwoPiHi); rol.add(-kTwoPiLo); } else if (rol.v < -kPiF) { rol.add(kTwoPiHi); rol.add(kTwoPiLo); }
        const float big = fmaxf(fabsf(dy), fmaxf(fabsf(dp), fabsf(dr)));
        if (do_resync || fold || !(big <= 0.25f)) {
            resync();
        } else {
            rotate(dy, sy, cy); rotate(dp, sp, cp); rotate(dr, sr, cr);
        }
    }
};

// (0,1) uniform on the grid (2k+1) 2^-24, k < 2^23, from the top 23 bits of a word: v_alignbit_b32 drops them under
// the exponent of 1.0f (a float in [1,2)), one subtraction moves it to (0,1).  Symmetric about 1/2; the smallest
// value 2^-24 bounds |z| at 5.77 sigma.
__device__ __forceinline__ float uniform23(uint32_t w) {
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, w, 9)) - 0x1.fffffep-1f;      // - (1 - 2^-24)
}

// four standard normals from one Philox block (two Box-Muller pairs on the hardware transcendental units:
// v_log_f32, v_sqrt_f32 -- 1 ulp, the argument -2 ln u is in [1.2e-7, 33.3] -- and v_sin_f32 / v_cos_f32)
__device__ __forceinline__ void normals4(const RngKey& key, uint32_t stream, uint32_t j, float (&z)[4]) {
    const u32x4 w = philox4x32(j, stream, key.r0, key.r1, key.k0, key.k1);
    const float u0 = uniform23(w.x), u1 = uniform23(w.y), u2 = uniform23(w.z), u3 = uniform23(w.w);
    // -2 ln u = -2 ln2 log2 u ; v_sin_f32 / v_cos_f32 take the angle in revolutions
    const float r0 = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(u0));
    const float r1 = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(u2));
    z[0] = r0 * __builtin_amdgcn_cosf(u1);
    z[1] = r0 * __builtin_amdgcn_sinf(u1);
    z[2] = r1 * __builtin_amdgcn_cosf(u3);
    z[3] = r1 * __builtin_amdgcn_sinf(u3);
}

// The twelve IMU normals of a step from TWO Philox blocks (256 bits): six Box-Muller pairs of a 23-bit radius uniform
// (top 23 bits of A0..A3, B0, B1) and an 18-bit angle uniform cut from B2, B3 and the spare low bits of the radius
// words.  z[2p] = r_p cos, z[2p+1] = r_p sin.
__device__ __forceinline__ void normals12(const RngKey& key, uint32_t j, float (&z)[12]) {
    const u32x4 A = philox4x32(j, 0u, key.r0, key.r1, key.k0, key.k1);
    const u32x4 B = philox4x32(j, 1u, key.r0, key.r1, key.k0, key.k1);
    const uint32_t rw[6] = {A.x, A.y, A.z, A.w, B.x, B.y};
    const uint32_t aw[6] = {B.z >> 14, B.w >> 14,
                            ((B.z & 0x3fffu) << 4) | (A.x & 0xfu), ((B.w & 0x3fffu) << 4) | (A.y & 0xfu),
                            ((A.z & 0x1ffu) << 9) | (A.w & 0x1ffu), ((B.x & 0x1ffu) << 9) | (B.y & 0x1ffu)};
    float rad[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) rad[p] = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(uniform23(rw[p])));
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const float rev = __builtin_fmaf((float)aw[p], 0x1.0p-18f, 0x1.0p-19f);      // (k + 1/2) 2^-18 revolutions
        z[2 * p] = rad[p] * __builtin_amdgcn_cosf(rev);
        z[2 * p + 1] = rad[p] * __builtin_amdgcn_sinf(rev);
    }
}

struct Nav {
    Att att;
    Acc vb[3];      // body velocity (ref_frame 1)
    Acc vn[3];      // NED velocity state (ref_frame 0)
    V3 vel;         // navigation-frame velocity of the previous sample
    double pos[3];  // absolute position, fp64
    double pos0[3];
    float g;
    float sl, cl;   // sin/cos latitude (ref_frame 0), refreshed every resync
    bool ext_g;
};

template <int RF>
__device__ __forceinline__ void nav_init(Nav& s, const double* __restrict__ ini, int has_g) {
    s.att.set((float)ini[6], (float)ini[7], (float)ini[8]);
    const V3 vb{(float)ini[3], (float)ini[4], (float)ini[5]};
    s.vb[0] = Acc{vb.x, 0.f}; s.vb[1] = Acc{vb.y, 0.f}; s.vb[2] = Acc{vb.z, 0.f};
    s.vel = s.att.to_nav(vb);
    s.vn[0] = Acc{s.vel.x, 0.f}; s.vn[1] = Acc{s.vel.y, 0.f}; s.vn[2] = Acc{s.vel.z, 0.f};
    if (RF == 1) {
        const Vec3 e = lla2ecef(ini[0], ini[1], ini[2]);
        s.pos[0] = e.x; s.pos[1] = e.y; s.pos[2] = e.z;
        s.g = (float)(has_g ? ini[9] : geo_param(ini[0], ini[2]).g);
    } else {
        s.pos[0] = ini[0]; s.pos[1] = ini[1]; s.pos[2] = ini[2];
        s.g = has_g ? (float)ini[9] : 0.f;
    }
    s.pos0[0] = s.pos[0]; s.pos0[1] = s.pos[1]; s.pos0[2] = s.pos[2];
    double sl, cl;
    sincos(ini[0], &sl, &cl);
    s.sl = (float)sl; s.cl = (float)cl;
    s.ext_g = has_g != 0;
}

template <int RF, bool ODO>
__device__ __forceinline__ void nav_step(Nav& s, const V3& gyro, const V3& accel, float odo, float dt, int earth_rot, bool resync) {
    if (RF == 1) {
        const V3 v_prev = s.vel;
        if (!ODO) {
            const V3 gb = s.att.down_in_body();
            const V3 vb{s.vb[0].v, s.vb[1].v, s.vb[2].v};
            const V3 wxv = cross(gyro, vb);
            s.vb[0].add((accel.x + gb.x * s.g - wxv.x) * dt);
            s.vb[1].add((accel.y + gb.y * s.g - wxv.y) * dt);
            s.vb[2].add((accel.z + gb.z * s.g - wxv.z) * dt);
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
        const float qq = 1.0f - (float)kEsq * s2;
        const float iw = __frsqrt_rn(qq);
        const float rn = (float)kRe * iw;
        const float rm = (float)(kRe * (1.0 - kEsq)) * iw * (iw * iw);
        const float gm = (float)kG0 * (1.0f + (float)kGk * s2) * iw *
                         (1.0f - (float)(2.0 / kRe) * (1.0f + (float)(kFlat + kGm) - 2.0f * (float)kFlat * s2) * h +
                          (float)(3.0 / (kRe * kRe)) * (h * h));
        const float irm = __builtin_amdgcn_rcpf(rm + h), irn = __builtin_amdgcn_rcpf(rn + h), icl = __builtin_amdgcn_rcpf(s.cl);
        const V3 v = s.vel;
        const V3 w_en{v.y * irn, -v.x * irm, -v.y * s.sl * icl * irn};
        V3 w_ie{0.f, 0.f, 0.f};
        if (earth_rot) { w_ie.x = (float)kWie * s.cl; w_ie.z = -(float)kWie * s.sl; }
        const V3 wb = s.att.to_body(V3{w_en.x + w_ie.x, w_en.y + w_ie.y, w_en.z + w_ie.z});
        const V3 w_nb{gyro.x - wb.x, gyro.y - wb.y, gyro.z - wb.z};
        if (!ODO) {
            const V3 an = s.att.to_nav(accel);
            const float g = s.ext_g ? s.g : gm;
            const V3 cor = cross(V3{2.f * w_ie.x + w_en.x, 2.f * w_ie.y + w_en.y, 2.f * w_ie.z + w_en.z}, v);
            s.vn[0].add((an.x - cor.x) * dt);
            s.vn[1].add((an.y - cor.y) * dt);
            s.vn[2].add((an.z + g - cor.z) * dt);
        }
        s.att.step(w_nb, dt, resync);
        const double dlat = (double)(v.x * irm * dt);
        s.pos[0] += dlat;
        s.pos[1] += (double)(v.y * irn * icl * dt);
        s.pos[2] += (double)(-v.z * dt);
        if (resync) {
            double sl, cl;
            sincos(s.pos[0], &sl, &cl);
            s.sl = (float)sl; s.cl = (float)cl;
        } else {
            rotate((float)dlat, s.sl, s.cl);
        }
        if (ODO) {
            const V3 f = s.att.fwd_in_nav();
            s.vel = V3{f.x * odo, f.y * odo, f.z * odo};
        } else {
            s.vel = V3{s.vn[0].v, s.vn[1].v, s.vn[2].v};
        }
    }
}

// written once, never read back by the kernel: non-temporal stores
__device__ __forceinline__ void st(float* p, float v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ void store9(float* __restrict__ base, int64_t plane, int64_t off, const Nav& s) {
    st(base + 0 * plane + off, s.att.yaw.v);
    st(base + 1 * plane + off, s.att.pit.v);
    st(base + 2 * plane + off, s.att.rol.v);
    st(base + 3 * plane + off, (float)(s.pos[0] - s.pos0[0]));      // displacement from the initial position
    st(base + 4 * plane + off, (float)(s.pos[1] - s.pos0[1]));
    st(base + 5 * plane + off, (float)(s.pos[2] - s.pos0[2]));
    st(base + 6 * plane + off, s.vel.x);
    st(base + 7 * plane + off, s.vel.y);
    st(base + 8 * plane + off, s.vel.z);
}

typedef const ginsim_mc_params __attribute__((address_space(4))) * params_ptr;
typedef const double __attribute__((address_space(4))) * uniform_ptr;

__device__ __forceinline__ void store_end(double* __restrict__ out, int64_t runs, int64_t r, const Nav& s) {
    params_ptr kp = (params_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    out[0 * runs + r] = angle_range_pi((double)s.att.yaw.v - kp->ref_end[0]);
    out[1 * runs + r] = angle_range_pi((double)s.att.pit.v - kp->ref_end[1]);
    out[2 * runs + r] = angle_range_pi((double)s.att.rol.v - kp->ref_end[2]);
    Vec3 ep{s.pos[0] - kp->ref_end[3], s.pos[1] - kp->ref_end[4], s.pos[2] - kp->ref_end[5]};
    if (kp->end_pos_ned && kp->ref_frame == 0)
        ep = lla_error_ned(Vec3{s.pos[0], s.pos[1], s.pos[2]}, Vec3{kp->ref_end[3], kp->ref_end[4], kp->ref_end[5]});
    out[3 * runs + r] = ep.x;
    out[4 * runs + r] = ep.y;
    out[5 * runs + r] = ep.z;
    out[6 * runs + r] = (double)s.vel.x - kp->ref_end[6];
    out[7 * runs + r] = (double)s.vel.y - kp->ref_end[7];
    out[8 * runs + r] = (double)s.vel.z - kp->ref_end[8];
}

struct Model { float bias[3], a[3], b[3], w[3]; int wd[3]; };

__device__ __forceinline__ void load_model(const ginsim_sensor_model& m, Model& o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.bias[i] = (float)m.bias[i]; o.a[i] = (float)m.gm_a[i]; o.b[i] = (float)m.gm_b[i]; o.w[i] = (float)m.white[i];
        o.wd[i] = m.white_drift[i];
    }
}

__device__ __forceinline__ V3 sense3(const double (&truth)[3], const Model& m, float (&drift)[3], const float* zd, const float* zw) {
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float bz = m.b[i] * zd[i];
        const float d = m.wd[i] ? bz : drift[i];
        o[i] = (float)truth[i] + m.bias[i] + d + m.w[i] * zw[i];
        drift[i] = fmaf(m.a[i], drift[i], bz);
    }
    return V3{o[0], o[1], o[2]};
}

template <int RF, int ALGOS>
__global__ void __launch_bounds__(256) mc_kernel_f32(const ginsim_mc_params a) {
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
    float* o_acc = reinterpret_cast<float*>(a.out_accel);
    float* o_gyr = reinterpret_cast<float*>(a.out_gyro);
    float* o_odo = reinterpret_cast<float*>(a.out_odo);
    float* o_fi = reinterpret_cast<float*>(a.out_traj[0]);
    float* o_od = reinterpret_cast<float*>(a.out_traj[1]);
    const uniform_ptr ref_a = (uniform_ptr)(uintptr_t)a.ref_accel, ref_g = (uniform_ptr)(uintptr_t)a.ref_gyro,
                      ref_o = (uniform_ptr)(uintptr_t)a.ref_odo;
    if (FREE && o_fi) store9(o_fi, plane, r, fi);
    if (ODO && o_od) store9(o_od, plane, r, od);
    for (int64_t j = 0; j < n; ++j) {
        const int64_t off = j * runs + r;
        const bool last = (j == n - 1);
        if (last && !o_acc && !o_gyr && !o_odo) break;
        const uint32_t jj = (uint32_t)j;
        // wave-uniform truth of this step, requested before the noise is generated (scalar-load latency hidden)
        const double ta[3] = {ref_a[3 * j], ref_a[3 * j + 1], ref_a[3 * j + 2]};
        const double tg[3] = {ref_g[3 * j], ref_g[3 * j + 1], ref_g[3 * j + 2]};
        // 12 normals from two Philox blocks: accel drift xyz, accel white xyz, gyro drift xyz, gyro white xyz
        float z[12];
        normals12(key, jj, z);
        const float zda[3] = {z[0], z[1], z[2]}, zwa[3] = {z[3], z[4], z[5]};
        const float zdg[3] = {z[6], z[7], z[8]}, zwg[3] = {z[9], z[10], z[11]};
        const V3 acc = sense3(ta, ma, da, zda, zwa);
        const V3 gyr = sense3(tg, mg, dg, zdg, zwg);
        if (o_acc) { st(o_acc + off, acc.x); st(o_acc + plane + off, acc.y); st(o_acc + 2 * plane + off, acc.z); }
        if (o_gyr) { st(o_gyr + off, gyr.x); st(o_gyr + plane + off, gyr.y); st(o_gyr + 2 * plane + off, gyr.z); }
        float odo = 0.f;
        if (ODO || o_odo) {
            float z3[4];
            normals4(key, S_ODO, jj, z3);
            odo = odo_scale * (float)ref_o[j] + odo_stdv * z3[0];
            if (o_odo) st(o_odo + off, odo);
        }
        if (last) break;
        const bool resync = ((j + 1) & (kTrigResync - 1)) == 0;
        if (FREE) {
            nav_step<RF, false>(fi, gyr, acc, 0.f, dt, a.earth_rot, resync);
            if (o_fi) store9(o_fi, plane, off + runs, fi);
        }
        if (ODO) {
            nav_step<RF, true>(od, gyr, acc, odo, dt, a.earth_rot, resync);
            if (o_od) store9(o_od, plane, off + runs, od);
        }
    }
    if (FREE && a.out_end[0]) store_end(a.out_end[0], runs, r, fi);
    if (ODO && a.out_end[1]) store_end(a.out_end[1], runs, r, od);
}

// Wave-specialised variant for batches of <= 1024 wavefronts (see mc_kernel_split in mc_kernel.hip for the rationale):
// waves 4-7 produce the twelve normals of a step (two Philox blocks, six Box-Muller pairs) into an LDS ring, waves 0-3
// consume them and do sensors, mechanisation and stores.  Bit-identical to mc_kernel_f32.
constexpr int kSplitTileF = 6;
constexpr int kSplitRunsF = 256;
constexpr size_t kSplitLdsF = sizeof(float) * 2 * kSplitTileF * 12 * kSplitRunsF;      // 144 KiB

template <int RF, int ALGOS>
__global__ void __launch_bounds__(512) mc_kernel_f32_split(const ginsim_mc_params a) {
    extern __shared__ float zringf[];
    constexpr bool FREE = (ALGOS & GINSIM_ALGO_FREE) != 0;
    constexpr bool ODO = (ALGOS & GINSIM_ALGO_ODO) != 0;
    const int lane = threadIdx.x & (kSplitRunsF - 1);
    const bool producer = threadIdx.x >= kSplitRunsF;
    const int64_t r = (int64_t)blockIdx.x * kSplitRunsF + lane;
    const bool active = r < a.runs;
    const int64_t n = a.n, runs = a.runs, plane = n * runs;
    float* o_acc = reinterpret_cast<float*>(a.out_accel);
    float* o_gyr = reinterpret_cast<float*>(a.out_gyro);
    float* o_odo = reinterpret_cast<float*>(a.out_odo);
    float* o_fi = reinterpret_cast<float*>(a.out_traj[0]);
    float* o_od = reinterpret_cast<float*>(a.out_traj[1]);
    const bool keep_last = o_acc || o_gyr || o_odo;
    const int64_t n_noise = keep_last ? n : n - 1;
    const int64_t ntiles = (n_noise + kSplitTileF - 1) / kSplitTileF;
    const uint64_t grun = a.run_offset + (uint64_t)r;
    const RngKey key{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)grun, (uint32_t)(grun >> 32)};

    if (producer) {
        for (int64_t i = 0; i <= ntiles; ++i) {
            if (i < ntiles && active) {
                float* zb = zringf + (i & 1) * (kSplitTileF * 12 * kSplitRunsF) + lane;
#pragma unroll
                for (int t = 0; t < kSplitTileF; ++t) {
                    const int64_t j = i * kSplitTileF + t;
                    if (j < n_noise) {
                        float z[12];
                        normals12(key, (uint32_t)j, z);
#pragma unroll
                        for (int k = 0; k < 12; ++k) zb[(t * 12 + k) * kSplitRunsF] = z[k];
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
    const uniform_ptr ref_a = (uniform_ptr)(uintptr_t)a.ref_accel, ref_g = (uniform_ptr)(uintptr_t)a.ref_gyro,
                      ref_o = (uniform_ptr)(uintptr_t)a.ref_odo;
    if (active) {
        if (FREE && o_fi) store9(o_fi, plane, r, fi);
        if (ODO && o_od) store9(o_od, plane, r, od);
    }
    for (int64_t i = 0; i <= ntiles; ++i) {
        if (i >= 1 && active) {
            const float* zb = zringf + ((i - 1) & 1) * (kSplitTileF * 12 * kSplitRunsF) + lane;
#pragma unroll 1
            for (int t = 0; t < kSplitTileF; ++t) {
                const int64_t j = (i - 1) * kSplitTileF + t;
                if (j >= n_noise) break;
                const int64_t off = j * runs + r;
                const bool last = (j == n - 1);
                const double ta[3] = {ref_a[3 * j], ref_a[3 * j + 1], ref_a[3 * j + 2]};
                const double tg[3] = {ref_g[3 * j], ref_g[3 * j + 1], ref_g[3 * j + 2]};
                float z[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) z[k] = zb[(t * 12 + k) * kSplitRunsF];
                const float zda[3] = {z[0], z[1], z[2]}, zwa[3] = {z[3], z[4], z[5]};
                const float zdg[3] = {z[6], z[7], z[8]}, zwg[3] = {z[9], z[10], z[11]};
                const V3 acc = sense3(ta, ma, da, zda, zwa);
                const V3 gyr = sense3(tg, mg, dg, zdg, zwg);
                if (o_acc) { st(o_acc + off, acc.x); st(o_acc + plane + off, acc.y); st(o_acc + 2 * plane + off, acc.z); }
                if (o_gyr) { st(o_gyr + off, gyr.x); st(o_gyr + plane + off, gyr.y); st(o_gyr + 2 * plane + off, gyr.z); }
                float odo = 0.f;
                if (ODO || o_odo) {
                    float z3[4];
                    normals4(key, S_ODO, (uint32_t)j, z3);
                    odo = odo_scale * (float)ref_o[j] + odo_stdv * z3[0];
                    if (o_odo) st(o_odo + off, odo);
                }
                if (last) break;
                const bool resync = ((j + 1) & (kTrigResync - 1)) == 0;
                if (FREE) {
                    nav_step<RF, false>(fi, gyr, acc, 0.f, dt, a.earth_rot, resync);
                    if (o_fi) store9(o_fi, plane, off + runs, fi);
                }
                if (ODO) {
                    nav_step<RF, true>(od, gyr, acc, odo, dt, a.earth_rot, resync);
                    if (o_od) store9(o_od, plane, off + runs, od);
                }
            }
        }
        __syncthreads();
    }
    if (active) {
        if (FREE && a.out_end[0]) store_end(a.out_end[0], runs, r, fi);
        if (ODO && a.out_end[1]) store_end(a.out_end[1], runs, r, od);
    }
}

}  // namespace f32

static int split_policy_f32() {
    static const int v = [] { const char* e = getenv("GINSIM_SPLIT"); return e ? atoi(e) : -1; }();
    return v;
}

int mc_variant_f32(const ginsim_mc_params& p) {
    if (!(p.algo_mask & GINSIM_ALGO_FREE) || p.n < 2) return 0;
    const int pol = split_policy_f32();
    return pol == 1 || (pol < 0 && (p.runs + 63) / 64 <= 1024);
}

template <int RF, int ALGOS>
static hipError_t launch2_f32(const ginsim_mc_params& p, hipStream_t stream) {
    const int tb = 256;
    const int64_t waves = (p.runs + 63) / 64;
    if constexpr ((ALGOS & GINSIM_ALGO_FREE) != 0) {
        if (mc_variant_f32(p)) {
            static bool once = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&f32::mc_kernel_f32_split<RF, ALGOS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)f32::kSplitLdsF);
                return true;
            }();
            (void)once;
            hipLaunchKernelGGL((f32::mc_kernel_f32_split<RF, ALGOS>), dim3((unsigned)((p.runs + 255) / 256)), dim3(512),
                               f32::kSplitLdsF, stream, p);
            return hipGetLastError();
        }
    }
    // exactly k workgroups per CU (k + 1 do not fit the LDS reservation): 148 VGPRs allow 3 wavefronts per SIMD
    const int per_cu = waves <= 1024 ? 1 : (waves <= 2048 ? 2 : 3);
    const size_t lds = (160 * 1024) / (per_cu + 1) + 1024;
    hipLaunchKernelGGL((f32::mc_kernel_f32<RF, ALGOS>), dim3((unsigned)((p.runs + tb - 1) / tb)), dim3(tb), lds, stream, p);
    return hipGetLastError();
}

template <int RF>
static hipError_t launch1_f32(const ginsim_mc_params& p, hipStream_t stream) {
    switch (p.algo_mask) {
        case GINSIM_ALGO_FREE: return launch2_f32<RF, GINSIM_ALGO_FREE>(p, stream);
        case GINSIM_ALGO_ODO: return launch2_f32<RF, GINSIM_ALGO_ODO>(p, stream);
        default: return launch2_f32<RF, GINSIM_ALGO_FREE | GINSIM_ALGO_ODO>(p, stream);
    }
}

hipError_t launch_mc_f32(const ginsim_mc_params& p, hipStream_t stream) {
    return p.ref_frame == 1 ? launch1_f32<1>(p, stream) : launch1_f32<0>(p, stream);
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
