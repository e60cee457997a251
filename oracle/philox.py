"""Counter-based normal generator of the MC engine, NumPy restatement (TEST ORACLE).

The reference draws its noise from the global legacy ``np.random.randn`` stream
(/root/reference/gnss_ins_sim/pathgen/pathgen.py:495,557,588,593,621-622,639,660)
which is serial and unseeded: "parity unpinned" for the stream itself.  The
engine therefore defines its own stream -- Philox4x32-7 (Salmon et al., SC'11,
"Parallel random numbers: as easy as 1, 2, 3": seven rounds is the Crush-resistant
Philox4x32 of the paper, ten its conservative default; Random123 v1.14 constants and
known-answer vectors for both round counts are checked in the tests) followed by
a Box-Muller transform that is DEFINED operation by operation in IEEE single precision
(every multiplication, addition and square root rounded separately, no fused multiply-adds,
two committed lookup tables) -- so the device, this file and the C oracle produce the SAME BITS,
and parity is "identical injected normals": the unmodified reference is fed THESE normals
through a ``np.random.randn`` shim (``oracle/ref_shim.py``).

Stream definition (shared by this file, ``oracle/c/ginsim_oracle.c`` and
``gnss-ins-sim_amd/csrc/philox.hpp`` / ``fastmath.hpp``).  A normal pair takes the two words of a half block, so one
128-bit block gives TWO pairs and the six pairs of an IMU step are exactly three blocks:

    key     = (seed & 0xffffffff, seed >> 32)
    W       = philox4x32_7((j, s >> 1, run & 0xffffffff, run >> 32), key)      j = sample index, s = stream id
    (a, b)  = (W0, W1) if s is even else (W2, W3)
    u   = (f32(a) + 0.5) * 2**-32                                   radius uniform in (0, 1]: |z| up to 6.7 sigma
    x   = -2 ln u        by exponent / 256-bin mantissa table / cubic   (radius2_f32 below)
    r   = sqrt(x)        correctly rounded
    s,c = sin, cos of 2 pi ((b & 0xffffff) + 1/2) 2**-24     by 512-sector table / two-term series (sincos_f32 below)
    z0 = f64(r * c) ;  z1 = f64(r * s)

Stream ids (one stream -> two normals (z0, z1)):

    0: accel drift x, y     1: accel drift z, accel white x    2: accel white y, z
    3: gyro  drift x, y     4: gyro  drift z, gyro  white x    5: gyro  white y, z
    6: odometer, -          7: mag x, y                        8: mag z, -
    15: gps pos x, y       16: gps pos z, vel x               17: gps vel y, z            (j = GPS sample index)
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
SQRT_HALF_BITS = np.uint32(0x3f3504f3)          # bits of f32(sqrt(1/2))
NEG_2LN2 = F32(-1.3862943611198906)
ANG_SCALE = F32(2.0 * np.pi * 2.0 ** -24)


def normal_tables():
    """The two fp32 tables of the generator -- the same construction as tools/gen_normal_tables.py, whose output
    (csrc/normal_tables.inc) the device and the C oracle use; tests/test_oracle_golden.py compares the bits."""
    k = np.arange(256, dtype=np.uint32)
    lo = (SQRT_HALF_BITS + (k << np.uint32(15))).view(F32).astype(np.float64)
    hi = (SQRT_HALF_BITS + (k << np.uint32(15)) + np.uint32(0x8000)).view(F32).astype(np.float64)
    unit = (lo <= 1.0) & (1.0 < hi)
    c = np.where(unit, 1.0, 0.5 * (lo + hi)).astype(F32)
    c64 = c.astype(np.float64)
    lg = np.zeros((256, 3), dtype=F32)
    lg[:, 0] = c
    lg[:, 1] = (-2.0 / c64).astype(F32)
    lg[:, 2] = np.where(unit, 0.0, -2.0 * np.log(c64)).astype(F32)
    ang = 2.0 * np.pi * (np.arange(512) + 0.5) / 512.0
    return lg, np.stack([np.sin(ang), np.cos(ang)], axis=1).astype(F32)


_LG, _SC = normal_tables()


def radius2_f32(a):
    """x = -2 ln u, u = (f32(a) + 1/2) 2^-32, in single precision: u = m 2^e with m in [sqrt(1/2), sqrt(2)),
    x = e (-2 ln 2) + (-2 ln c_k) + (r + r^2 (1/4 + r/12)),  r = (m - c_k)(-2 / c_k).  Every operation is one IEEE op."""
    t = np.asarray(a, dtype=np.uint64).astype(F32)                  # uint32 -> f32, round to nearest even
    u = (t + F32(0.5)) * F32(2.0 ** -32)
    hx = u.view(np.uint32) + (np.uint32(0x3f800000) - SQRT_HALF_BITS)
    ef = ((hx >> np.uint32(23)).astype(np.int32) - 127).astype(F32)
    tab = _LG[(hx >> np.uint32(15)) & np.uint32(255)]
    m = ((hx & np.uint32(0x007fffff)) + SQRT_HALF_BITS).view(F32)
    d = m - tab[..., 0]
    r = d * tab[..., 1]
    q = r * F32(1.0 / 12.0)
    q = q + F32(0.25)
    q = q * (r * r)
    small = r + q
    x = ef * NEG_2LN2
    x = x + tab[..., 2]
    return x + small


def sincos_f32(b):
    """sin, cos of 2 pi ((b & 0xffffff) + 1/2) 2^-24: sector = top 9 of the 24 bits, remainder centred, in radians."""
    w = np.asarray(b, dtype=np.uint64).astype(np.uint32)
    tab = _SC[(w >> np.uint32(15)) & np.uint32(511)]
    sn_i, cs_i = tab[..., 0], tab[..., 1]
    bb = (w & np.uint32(0x7fff)).astype(F32) + F32(0.5 - 16384.0)
    bb = bb * ANG_SCALE
    tt = bb * bb
    u1 = tt * F32(-1.0 / 6.0)
    u1 = u1 * bb
    sb = bb + u1                                                     # sin b
    cm = tt * F32(-0.5)                                              # cos b - 1
    p1 = cs_i * sb
    p1 = p1 + sn_i * cm
    q1 = cs_i * cm
    q1 = q1 - sn_i * sb
    return sn_i + p1, cs_i + q1


def stream_words(seed, run, stream, j):
    """The two words (a, b) of one stream at sample(s) j: half (stream & 1) of block (j, stream >> 1)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    run = np.asarray(run, dtype=np.uint64)
    jj = np.asarray(j, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    w = philox4x32(jj, np.uint64(int(stream) >> 1), run & MASK32, run >> np.uint64(32), k0, k1)
    return (w[2], w[3]) if int(stream) & 1 else (w[0], w[1])


def box_muller(a, b):
    """(z0, z1) from the two words of a half block; the values are exact single-precision numbers held in float64."""
    with np.errstate(all='ignore'):
        r = np.sqrt(radius2_f32(a))                                   # float32 sqrt: correctly rounded
        sn, cs = sincos_f32(b)
        assert r.dtype == F32 and sn.dtype == F32
        return (r * cs).astype(np.float64), (r * sn).astype(np.float64)


def normal_pair(seed, run, stream, j):
    """Two standard normals (z0, z1) for (run, stream, sample j); arrays broadcast."""
    return box_muller(*stream_words(seed, run, stream, j))


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
