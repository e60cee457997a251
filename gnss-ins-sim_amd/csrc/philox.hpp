// Counter-based normal generator of the MC engine (device side).
//
// Philox4x32-7 (Salmon, Moraes, Dror, Shaw, SC'11: seven rounds is the paper's Crush-resistant Philox4x32, ten its
// conservative default; Random123 constants and known-answer vectors for both round counts are in the tests) +
// a normal transform defined bit-exactly in single precision (fastmath.hpp normal_icdf: one 32-bit word -> one normal by
// piecewise-cubic inversion of the tail probability, |z| up to 6.23 sigma).  A stream's pair of normals takes the two
// words of a half block, so one 128-bit block yields TWO pairs (four normals), and the six pairs (twelve normals) of an
// IMU step cost exactly three blocks:
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
    // vibration (Sim(env=...)): the three 'random' normals of a sample are one block per sensor; the three phase uniforms of a
    // 'sinusoidal' vibration are words 0..2 of ONE block per run and sensor (sample 0), u = word * 2^-32
    S_ACC_VIB_XY = 10, S_ACC_VIB_Z = 11, S_GYR_VIB_XY = 12, S_GYR_VIB_Z = 13,
    S_GYR_VIB_PHASE = 24, S_ACC_VIB_PHASE = 26,
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

struct RngKey {
    uint32_t k0, k1;    // seed
    uint32_t r0, r1;    // global run id
};

// N (word a, word b) draws -> N pairs of normals: z0 = normal_icdf(a), z1 = normal_icdf(b) (fastmath.hpp).  The normals are
// single-precision numbers.
template <int N>
__device__ __forceinline__ void normal_transform(const uint32_t (&a)[N], const uint32_t (&b)[N], float (&z0)[N], float (&z1)[N],
                                                 const NormalTables& tab) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        z0[k] = normal_icdf(a[k], tab);
        z1[k] = normal_icdf(b[k], tab);
    }
}

// The two words (a, b) of the N consecutive streams FIRST .. FIRST+N-1
// at sample j: blocks FIRST >> 1 .. (FIRST+N-1) >> 1, each computed once.
template <uint32_t FIRST, int N>
__device__ __forceinline__ void draw_streams(const RngKey& key, uint32_t j, uint32_t* a, uint32_t* b) {
    constexpr uint32_t B0 = FIRST >> 1, B1 = (FIRST + N - 1) >> 1;
#pragma unroll
    for (uint32_t blk = B0; blk <= B1; ++blk) {
        const u32x4 w = philox4x32(j, blk, key.r0, key.r1, key.k0, key.k1);
        if (2 * blk >= FIRST) {
            a[2 * blk - FIRST] = w.x;
            b[2 * blk - FIRST] = w.y;
        }
        if (2 * blk + 1 < FIRST + N) {
            a[2 * blk + 1 - FIRST] = w.z;
            b[2 * blk + 1 - FIRST] = w.w;
        }
    }
}

// N consecutive streams FIRST .. FIRST+N-1 of one sample -> N normal pairs, as single-precision numbers ...
template <uint32_t FIRST, int N>
__device__ __forceinline__ void normal_pairs_f32(const RngKey& key, uint32_t j, float (&z0)[N], float (&z1)[N], const NormalTables& tab) {
    uint32_t a[N], b[N];
    draw_streams<FIRST, N>(key, j, a, b);
    normal_transform<N>(a, b, z0, z1, tab);
}

// ... and widened to fp64 (exact) for the fp64 sensor models
template <uint32_t FIRST, int N>
__device__ __forceinline__ void normal_pairs(const RngKey& key, uint32_t j, double (&z0)[N], double (&z1)[N], const NormalTables& tab) {
    float f0[N], f1[N];
    normal_pairs_f32<FIRST, N>(key, j, f0, f1, tab);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        z0[k] = (double)f0[k];
        z1[k] = (double)f1[k];
    }
}

// Two standard normals of one stream (run-time stream id).
__device__ __forceinline__ void normal_pair(const RngKey& key, uint32_t stream, uint32_t j, double& z0, double& z1,
                                            const NormalTables& tab) {
    const u32x4 w = philox4x32(j, stream >> 1, key.r0, key.r1, key.k0, key.k1);
    const bool hi = (stream & 1u) != 0;
    const uint32_t a[1] = {hi ? w.z : w.x}, b[1] = {hi ? w.w : w.y};
    float f0[1], f1[1];
    normal_transform<1>(a, b, f0, f1, tab);
    z0 = (double)f0[0];
    z1 = (double)f1[0];
}

}  // namespace ginsim
