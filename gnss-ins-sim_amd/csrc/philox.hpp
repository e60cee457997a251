// Counter-based normal generator of the MC engine (device side).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123 constants) + Box-Muller in fp64.  A pair of normals
// takes a 53-bit radius uniform and a 32-bit angle uniform; three pairs (streams 3g, 3g+1, 3g+2) are cut from the 256
// bits of the two blocks (j, 2g) and (j, 2g+1) -- the cut is spelled out in oracle/philox.py.
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
    S_GPS_P_XY = 15, S_GPS_PZ_VX = 16, S_GPS_V_YZ = 17,
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
// of the 2^53 values rounds to 1.0).  top 2^-21 + (low 2^-53 + 2^-54): the inner FMA is exact (33 bits), the outer one
// rounds the exact sum once -- the bits of (v + 0.5) 2^-53 in two FMAs on three register constants.
__device__ __forceinline__ double uniform53(uint32_t lo, uint32_t hi, const MathConsts& k) {
    const uint32_t top = hi >> 11;                         // 21 bits
    const uint32_t low = (hi << 21) | (lo >> 11);          // 32 bits (one v_alignbit_b32)
    return __builtin_fma((double)top, k.u_hi, __builtin_fma((double)low, k.u_lo, k.u_half));
}

struct RngKey {
    uint32_t k0, k1;    // seed
    uint32_t r0, r1;    // global run id
};

// The three streams of group g at sample j: radius uniforms u[3] and angle words ang[3] from two Philox blocks.
__device__ __forceinline__ void draw_group(const RngKey& key, uint32_t g, uint32_t j, double* u, uint32_t* ang,
                                           const MathConsts& mk) {
    const u32x4 A = philox4x32_10(j, 2 * g, key.r0, key.r1, key.k0, key.k1);
    const u32x4 B = philox4x32_10(j, 2 * g + 1, key.r0, key.r1, key.k0, key.k1);
    u[0] = uniform53(A.x, A.y, mk);
    ang[0] = A.z;
    u[1] = uniform53(A.w, B.x, mk);
    ang[1] = B.y;
    u[2] = uniform53(((A.x & 0x7ffu) << 21) | ((A.w & 0x7ffu) << 10), B.z, mk);     // the spare low bits of A.x and A.w
    ang[2] = B.w;
}

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
        sincos_turn32(ang[k], s, c, mk, tab);
        z0[k] = r[k] * c;
        z1[k] = r[k] * s;
    }
}

// N consecutive streams first .. first+N-1 of one sample (whole groups: first and N multiples of 3) -> N normal pairs
template <int N>
__device__ __forceinline__ void normal_pairs(const RngKey& key, uint32_t first, uint32_t j, double (&z0)[N], double (&z1)[N],
                                             const MathConsts& mk, const NormalTables& tab) {
    static_assert(N % 3 == 0, "streams come in groups of three");
    double r[N];
    uint32_t ang[N];
#pragma unroll
    for (int g = 0; g < N / 3; ++g) draw_group(key, first / 3 + g, j, r + 3 * g, ang + 3 * g, mk);
    box_muller<N>(r, ang, z0, z1, mk, tab);
}

// Two standard normals of one stream.  Slot 0 of a group needs the first block only.
__device__ __forceinline__ void normal_pair(const RngKey& key, uint32_t stream, uint32_t j, double& z0, double& z1,
                                            const MathConsts& mk, const NormalTables& tab) {
    const uint32_t g = stream / 3, slot = stream - 3 * g;
    double r[1], a[1], b[1];
    uint32_t ang[1];
    if (slot == 0) {
        const u32x4 A = philox4x32_10(j, 2 * g, key.r0, key.r1, key.k0, key.k1);
        r[0] = uniform53(A.x, A.y, mk);
        ang[0] = A.z;
    } else {
        double u[3];
        uint32_t w[3];
        draw_group(key, g, j, u, w, mk);
        r[0] = slot == 1 ? u[1] : u[2];
        ang[0] = slot == 1 ? w[1] : w[2];
    }
    box_muller<1>(r, ang, a, b, mk, tab);
    z0 = a[0];
    z1 = b[0];
}

}  // namespace ginsim
