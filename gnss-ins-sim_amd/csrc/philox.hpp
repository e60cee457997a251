// Counter-based normal generator of the MC engine (device side).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123 constants) + Box-Muller in fp64.
// It replaces the reference's serial global np.random.randn stream
// (gnss_ins_sim/pathgen/pathgen.py:495,557,588,593,621-622,639,660): every (run, stream, sample)
// triple owns its variates, so lanes never share RNG state and only variates that are consumed are
// generated.  The stream definition is restated in oracle/philox.py and must stay in lock-step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fastmath.hpp"

namespace ginsim {

enum : uint32_t {
    S_ACC_D_XY = 0, S_ACC_DZ_WX = 1, S_ACC_W_YZ = 2,
    S_GYR_D_XY = 3, S_GYR_DZ_WX = 4, S_GYR_W_YZ = 5,
    S_ODO = 6, S_MAG_XY = 7, S_MAG_Z = 8,
    S_GPS_P_XY = 16, S_GPS_PZ_VX = 17, S_GPS_V_YZ = 18,
};

struct u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c in one v_bitop3_b32 (truth table 0x96); the compiler emits two v_xor_b32 when c is a scalar register
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
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

// (0,1] uniform with 53 significant bits from two words: u = ((hi:lo >> 11) + 0.5) * 2^-53, rounded once (the largest
// of the 2^53 values rounds to 1.0).  v + 0.5 rounds exactly like the scaled sum and the power of two is exact, so
// add + ldexp (two inline constants) gives the bits of fma(v, 2^-53, 2^-54) without materialising the constants.
__device__ __forceinline__ double uniform53(uint32_t lo, uint32_t hi) {
    const uint32_t top = hi >> 11;                         // 21 bits
    const uint32_t low = (hi << 21) | (lo >> 11);          // 32 bits (one v_alignbit_b32)
    const double v = __builtin_fma((double)top, 4294967296.0, (double)low);     // exact, < 2^53
    return __builtin_amdgcn_ldexp(v + 0.5, -53);
}

struct RngKey {
    uint32_t k0, k1;    // seed
    uint32_t r0, r1;    // global run id
};

// N consecutive streams (first, first+1, ...) of one sample -> N pairs of standard normals
//   z0 = sqrt(-2 ln u1) cos(2 pi u2),  z1 = sqrt(-2 ln u1) sin(2 pi u2),  u1 = uniform53(w.x, w.y), u2 = uniform53(w.z, w.w)
// evaluated phase by phase -- all Philox blocks, then all logarithms, then all square roots, then all sin/cos --
// instead of N complete Box-Muller transforms in a row.  Each phase is N independent dependency chains (ILP for a
// lone wavefront on its SIMD) and only ONE polynomial's constants are live at a time.  The angle never becomes a
// uniform: sincos_turn53 works on the integer.
template <int N>
__device__ __forceinline__ void normal_pairs(const RngKey& key, uint32_t first, uint32_t j, double (&z0)[N], double (&z1)[N],
                                             const MathConsts& mk, const NormalTables& tab) {
    double r[N];
    uint32_t a_lo[N], a_hi[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const u32x4 w = philox4x32_10(j, first + k, key.r0, key.r1, key.k0, key.k1);
        r[k] = uniform53(w.x, w.y);
        a_lo[k] = w.z;
        a_hi[k] = w.w;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = -2.0 * log_u01(r[k], mk, tab);
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = sqrt_pos(r[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s, c;
        sincos_turn53(a_lo[k], a_hi[k], s, c, mk, tab);
        z0[k] = r[k] * c;
        z1[k] = r[k] * s;
    }
}

// Two standard normals for (key.run, stream, sample j).
__device__ __forceinline__ void normal_pair(const RngKey& key, uint32_t stream, uint32_t j, double& z0, double& z1,
                                            const MathConsts& mk, const NormalTables& tab) {
    double a[1], b[1];
    normal_pairs<1>(key, stream, j, a, b, mk, tab);
    z0 = a[0];
    z1 = b[0];
}

}  // namespace ginsim
