// Counter-based normal generator of the MC engine (device side).
//
// Philox4x32-7 (Salmon, Moraes, Dror, Shaw, SC'11: seven rounds is the paper's Crush-resistant Philox4x32, ten its
// conservative default; Random123 constants and known-answer vectors for both round counts are in the tests) +
// Box-Muller in fp64.  A pair of normals takes 64 bits -- a 40-bit radius uniform (|z| up to 7.5 sigma) and a 24-bit
// angle uniform -- so one 128-bit block yields TWO pairs, and the six pairs (twelve normals) of an IMU step cost
// exactly three blocks:
//     stream s at sample j  =  half (s & 1) of block  philox4x32_7(counter = (j, s >> 1, run_lo, run_hi), key = seed)
// It replaces the reference's serial global np.random.randn stream
// (gnss_ins_sim/pathgen/pathgen.py:495,557,588,593,621-622,639,660): every (run, stream, sample)
// triple owns its variates, so lanes never share RNG state and only variates that are consumed are
// generated.  The stream definition is restated in oracle/philox.py and must stay in lock-step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fastmath.hpp"

namespace ginsim {

// stream ids: one stream -> two normals (z0, z1) per sample
enum : uint32_t {
    S_ACC_D_XY = 0, S_ACC_DZ_WX = 1, S_ACC_W_YZ = 2,
    S_GYR_D_XY = 3, S_GYR_DZ_WX = 4, S_GYR_W_YZ = 5,
    S_ODO = 6, S_MAG_XY = 7, S_MAG_Z = 8,
    S_GPS_P_XY = 15, S_GPS_PZ_VX = 16, S_GPS_V_YZ = 17,
};

struct u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c in one v_bitop3_b32 (truth table 0x96); the compiler emits two v_xor_b32 when c is a scalar register
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

constexpr int kPhiloxRounds = 7;

template <int ROUNDS = kPhiloxRounds>
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// (0,1) uniform with 40 significant bits from the two words (a, b) of a half block: u = ((a << 8 | b >> 24) + 0.5) 2^-40,
// exact in fp64: a 2^-32 + ((b >> 24) 2^-40 + 2^-41) as two FMAs on three register constants.
__device__ __forceinline__ double uniform40(uint32_t a, uint32_t b, const MathConsts& k) {
    return __builtin_fma((double)a, k.u_hi, __builtin_fma((double)(b >> 24), k.u_lo, k.u_half));
}

struct RngKey {
    uint32_t k0, k1;    // seed
    uint32_t r0, r1;    // global run id
};

// Box-Muller on N (radius uniform, angle word) draws, phase by phase -- all logarithms, then all square roots, then
// all sin/cos -- instead of N complete transforms in a row: each phase is N independent dependency chains (ILP for
// a lone wavefront on its SIMD) and only ONE polynomial's constants are live at a time.
template <int N>
__device__ __forceinline__ void box_muller(double (&r)[N], const uint32_t (&ang)[N], double (&z0)[N], double (&z1)[N],
                                           const MathConsts& mk, const NormalTables& tab) {
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = neg2_log_u01(r[k], mk, tab);
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = sqrt_pos(r[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s, c;
        sincos_turn24(ang[k], s, c, mk, tab);
        z0[k] = r[k] * c;
        z1[k] = r[k] * s;
    }
}

// Radius uniforms u[N] and angle words ang[N] (the low 24 bits count) of the N consecutive streams FIRST .. FIRST+N-1
// at sample j: blocks FIRST >> 1 .. (FIRST+N-1) >> 1, each computed once.
template <uint32_t FIRST, int N>
__device__ __forceinline__ void draw_streams(const RngKey& key, uint32_t j, double* u, uint32_t* ang, const MathConsts& mk) {
    constexpr uint32_t B0 = FIRST >> 1, B1 = (FIRST + N - 1) >> 1;
#pragma unroll
    for (uint32_t b = B0; b <= B1; ++b) {
        const u32x4 w = philox4x32(j, b, key.r0, key.r1, key.k0, key.k1);
        if (2 * b >= FIRST) {
            u[2 * b - FIRST] = uniform40(w.x, w.y, mk);
            ang[2 * b - FIRST] = w.y;
        }
        if (2 * b + 1 < FIRST + N) {
            u[2 * b + 1 - FIRST] = uniform40(w.z, w.w, mk);
            ang[2 * b + 1 - FIRST] = w.w;
        }
    }
}

// N consecutive streams FIRST .. FIRST+N-1 of one sample -> N normal pairs
template <uint32_t FIRST, int N>
__device__ __forceinline__ void normal_pairs(const RngKey& key, uint32_t j, double (&z0)[N], double (&z1)[N],
                                             const MathConsts& mk, const NormalTables& tab) {
    double r[N];
    uint32_t ang[N];
    draw_streams<FIRST, N>(key, j, r, ang, mk);
    box_muller<N>(r, ang, z0, z1, mk, tab);
}

// Two standard normals of one stream (run-time stream id).
__device__ __forceinline__ void normal_pair(const RngKey& key, uint32_t stream, uint32_t j, double& z0, double& z1,
                                            const MathConsts& mk, const NormalTables& tab) {
    const u32x4 w = philox4x32(j, stream >> 1, key.r0, key.r1, key.k0, key.k1);
    const bool hi = (stream & 1u) != 0;
    double r[1] = {uniform40(hi ? w.z : w.x, hi ? w.w : w.y, mk)}, a[1], b[1];
    const uint32_t ang[1] = {hi ? w.w : w.y};
    box_muller<1>(r, ang, a, b, mk, tab);
    z0 = a[0];
    z1 = b[0];
}

}  // namespace ginsim
