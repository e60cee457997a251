"""Counter-based normal generator of the MC engine, NumPy restatement (TEST ORACLE).

The reference draws its noise from the global legacy ``np.random.randn`` stream
(/root/reference/gnss_ins_sim/pathgen/pathgen.py:495,557,588,593,621-622,639,660)
which is serial and unseeded: "parity unpinned" for the stream itself.  The
engine therefore defines its own stream -- Philox4x32-7 (Salmon et al., SC'11,
"Parallel random numbers: as easy as 1, 2, 3": seven rounds is the Crush-resistant
Philox4x32 of the paper, ten its conservative default; Random123 v1.14 constants and
known-answer vectors for both round counts are checked in the tests) followed by
a normal transform that is DEFINED operation by operation in IEEE single precision
(integer bit manipulation, one row of a committed coefficient table, three fused multiply-adds)
-- so the device, this file and the C oracle produce the SAME BITS,
and parity is "identical injected normals": the unmodified reference is fed THESE normals
through a ``np.random.randn`` shim (``oracle/ref_shim.py``).

Stream definition (shared by this file, ``oracle/c/ginsim_oracle.c`` and
``gnss-ins-sim_amd/csrc/philox.hpp`` / ``fastmath.hpp``; the FOURTH one, round 3: one word per normal by inversion
instead of the single-precision Box-Muller of round 2 -- 13 ns instead of 30 ns of SIMD time per normal).  A stream's
pair of normals takes the two words of a half block, so one 128-bit block gives TWO pairs and the six pairs of an IMU
step are exactly three blocks:

    key     = (seed & 0xffffffff, seed >> 32)
    W       = philox4x32_7((j, s >> 1, run & 0xffffffff, run >> 32), key)      j = sample index, s = stream id
    (a, b)  = (W0, W1) if s is even else (W2, W3)
    z0 = f64(normal_icdf(a)) ;  z1 = f64(normal_icdf(b))
    normal_icdf(w):  m = (w & 0x7fffffff) | 1          upper tail probability t = m 2^-32 in (0, 1/2): |z| <= 6.23
                     lz = clz32(m);  y = m << lz       octave of t, then its eighth: seg = (lz - 1) 8 + ((y >> 28) & 7)
                     x = f32 bits 0x3f800000 | ((y << 4) >> 9)          in [1, 2)
                     z = fma(fma(fma(c3, x, c2), x, c1), x, c0)         (c0..c3) = table[seg], csrc/normal_tables.inc
                     |z| with the sign bit of w

Stream ids (one stream -> two normals (z0, z1)):

    0: accel drift x, y     1: accel drift z, accel white x    2: accel white y, z
    3: gyro  drift x, y     4: gyro  drift z, gyro  white x    5: gyro  white y, z
    6: odometer, -          7: mag x, y                        8: mag z, -
    15: gps pos x, y       16: gps pos z, vel x               17: gps vel y, z            (j = GPS sample index)
    10: accel vib x, y     11: accel vib z, -                 12: gyro vib x, y           13: gyro vib z, -
    ('random' vibration, Sim(env=...): one block per sensor and sample)
    24 / 26: the three phase uniforms of a 'sinusoidal' gyro / accel vibration = words W0, W1, W2 of block 12 / 13 at
    j = 0, u = W 2^-32 (one block per run and sensor)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

# stream ids
S_ACC_D_XY, S_ACC_DZ_WX, S_ACC_W_YZ = 0, 1, 2
S_GYR_D_XY, S_GYR_DZ_WX, S_GYR_W_YZ = 3, 4, 5
S_ODO, S_MAG_XY, S_MAG_Z = 6, 7, 8
S_GPS_P_XY, S_GPS_PZ_VX, S_GPS_V_YZ = 15, 16, 17
S_ACC_VIB_XY, S_ACC_VIB_Z, S_GYR_VIB_XY, S_GYR_VIB_Z = 10, 11, 12, 13
S_GYR_VIB_PHASE, S_ACC_VIB_PHASE = 24, 26


ROUNDS = 7


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=ROUNDS):
    """Philox4x32 with `rounds` rounds (7: the engine's generator; 10: Random123's default, kept for its
    known-answer vectors).  All arguments broadcastable uint32-valued arrays.

    Returns four uint64 arrays holding 32-bit words.
    """
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for rnd in range(rounds):
        p0 = M0 * c0            # 32x32 -> 64, cannot overflow uint64
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


F32 = np.float32
N_OCT, N_SUB = 31, 8


def normal_tables():
    """The coefficient table of the generator: the committed numbers (csrc/normal_tables.inc, made by
    tools/gen_normal_tables.py), which the device and the C oracle use as they are.  (248, 4) float32."""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gnss-ins-sim_amd', 'csrc', 'normal_tables.inc')
    words = [int(t, 16) for t in re.findall(r'0x([0-9a-fA-F]{8})u', open(path).read())]
    return np.array(words, dtype=np.uint32).view(F32).reshape(N_OCT * N_SUB, 4)


_Q = normal_tables()


def normal_icdf(w):
    """One standard normal from one 32-bit word by inversion of the upper tail probability t = m 2^-32, m = (w & 0x7fffffff) | 1:
    octave lz = clz(m), eighth of the octave = the 3 bits below the leading one, x in [1, 2) = the next 23 bits,
    z = c0 + x (c1 + x (c2 + x c3)) evaluated as three fused multiply-adds in single precision, sign = bit 31 of w.
    The fused multiply-add is emulated exactly (_fma32)."""
    w = np.asarray(w, dtype=np.uint64).astype(np.uint32)
    m = (w & np.uint32(0x7fffffff)) | np.uint32(1)
    nbits = np.floor(np.log2(m.astype(np.float64))).astype(np.int64) + 1            # bit length of m: 1 .. 31 (exact: m < 2^53)
    lz = 32 - nbits
    y = ((m.astype(np.uint64) << lz.astype(np.uint64)) & np.uint64(0xffffffff)).astype(np.uint32)
    c = _Q[(lz - 1) * N_SUB + ((y >> np.uint32(28)) & np.uint32(7)).astype(np.int64)]
    x = (np.uint32(0x3f800000) | ((y << np.uint32(4)) >> np.uint32(9))).view(F32)
    x64 = x.astype(np.float64)
    z = c[..., 3]
    for k in (2, 1, 0):
        z = _fma32(z, x64, c[..., k])
    zb = (z.view(np.uint32) & np.uint32(0x7fffffff)) | (w & np.uint32(0x80000000))
    return zb.view(F32)


LD = np.longdouble
assert np.finfo(LD).nmant >= 63, 'the fused multiply-add emulation needs an extended-precision long double (x86-64)'


def _fma32(a32, x64, c32):
    """float32(a * x + c) with ONE rounding: the product of two floats is exact in float64 (48 bits); its sum with a float of
    comparable size (the Horner terms here differ by < 2^16) is exact in the 64-bit significand of the x87 long double; the
    conversion to float32 then rounds once."""
    p = (a32.astype(np.float64) * x64).astype(LD)
    return (p + c32.astype(LD)).astype(F32)


def stream_words(seed, run, stream, j):
    """The two words (a, b) of one stream at sample(s) j: half (stream & 1) of block (j, stream >> 1)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    run = np.asarray(run, dtype=np.uint64)
    jj = np.asarray(j, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    w = philox4x32(jj, np.uint64(int(stream) >> 1), run & MASK32, run >> np.uint64(32), k0, k1)
    return (w[2], w[3]) if int(stream) & 1 else (w[0], w[1])


def normal_transform(a, b):
    """(z0, z1) from the two words of a half block: z0 = normal_icdf(a), z1 = normal_icdf(b); the values are exact
    single-precision numbers held in float64."""
    return normal_icdf(a).astype(np.float64), normal_icdf(b).astype(np.float64)


def normal_pair(seed, run, stream, j):
    """Two standard normals (z0, z1) for (run, stream, sample j); arrays broadcast."""
    return normal_transform(*stream_words(seed, run, stream, j))


def imu_normals(seed, run, n):
    """All IMU normals of one run.

    Returns dict of (n,3) arrays: 'acc_d', 'acc_w', 'gyr_d', 'gyr_w'.
    """
    j = np.arange(n, dtype=np.uint64)
    z = [normal_pair(seed, run, s, j) for s in range(6)]
    return {
        'acc_d': np.stack([z[0][0], z[0][1], z[1][0]], axis=1),
        'acc_w': np.stack([z[1][1], z[2][0], z[2][1]], axis=1),
        'gyr_d': np.stack([z[3][0], z[3][1], z[4][0]], axis=1),
        'gyr_w': np.stack([z[4][1], z[5][0], z[5][1]], axis=1),
    }


def odo_normals(seed, run, n):
    return normal_pair(seed, run, S_ODO, np.arange(n, dtype=np.uint64))[0]


def mag_normals(seed, run, n):
    j = np.arange(n, dtype=np.uint64)
    a = normal_pair(seed, run, S_MAG_XY, j)
    b = normal_pair(seed, run, S_MAG_Z, j)
    return np.stack([a[0], a[1], b[0]], axis=1)


def gps_normals(seed, run, m):
    j = np.arange(m, dtype=np.uint64)
    a = normal_pair(seed, run, S_GPS_P_XY, j)
    b = normal_pair(seed, run, S_GPS_PZ_VX, j)
    c = normal_pair(seed, run, S_GPS_V_YZ, j)
    return (np.stack([a[0], a[1], b[0]], axis=1),
            np.stack([b[1], c[0], c[1]], axis=1))


def vib_normals(seed, run, n, sensor):
    """(n,3) normals of a 'random' vibration (pathgen.py:485-488, 547-550); sensor 'acc' | 'gyr'."""
    j = np.arange(n, dtype=np.uint64)
    sxy, sz = (S_ACC_VIB_XY, S_ACC_VIB_Z) if sensor == 'acc' else (S_GYR_VIB_XY, S_GYR_VIB_Z)
    a = normal_pair(seed, run, sxy, j)
    b = normal_pair(seed, run, sz, j)
    return np.stack([a[0], a[1], b[0]], axis=1)


def vib_phase_uniforms(seed, run, sensor):
    """(3,) uniforms in [0, 1) of a 'sinusoidal' vibration's random phases (np.random.rand(1) x 3, pathgen.py:553-555):
    words 0..2 of the phase block at sample 0, times 2^-32."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    stream = S_ACC_VIB_PHASE if sensor == 'acc' else S_GYR_VIB_PHASE
    run = np.uint64(run)
    w = philox4x32(np.uint64(0), np.uint64(stream >> 1), run & MASK32, run >> np.uint64(32), seed & 0xFFFFFFFF, seed >> 32)
    return np.array([float(w[0]), float(w[1]), float(w[2])]) * 2.0 ** -32
