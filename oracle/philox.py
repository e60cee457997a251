"""Counter-based normal generator of the MC engine, NumPy restatement (TEST ORACLE).

The reference draws its noise from the global legacy ``np.random.randn`` stream
(/root/reference/gnss_ins_sim/pathgen/pathgen.py:495,557,588,593,621-622,639,660)
which is serial and unseeded: "parity unpinned" for the stream itself.  The
engine therefore defines its own stream -- Philox4x32-7 (Salmon et al., SC'11,
"Parallel random numbers: as easy as 1, 2, 3": seven rounds is the Crush-resistant
Philox4x32 of the paper, ten its conservative default; Random123 v1.14 constants and
known-answer vectors for both round counts are checked in the tests) followed by
Box-Muller in fp64 -- and parity is "identical injected normals": the
unmodified reference is fed THESE normals through a ``np.random.randn`` shim
(``oracle/ref_shim.py``).

Stream definition (shared by this file, ``oracle/c/ginsim_oracle.c`` and
``gnss-ins-sim_amd/csrc/philox.hpp``).  A normal pair takes 64 bits: a 40-bit uniform for the radius
(|z| up to 7.5 sigma) and a 24-bit uniform for the angle, so one 128-bit block gives TWO pairs and the six pairs
of an IMU step are exactly three blocks:

    key     = (seed & 0xffffffff, seed >> 32)
    W       = philox4x32_7((j, s >> 1, run & 0xffffffff, run >> 32), key)      j = sample index, s = stream id
    (a, b)  = (W0, W1) if s is even else (W2, W3)
    u1 = ((a << 8 | b >> 24) + 0.5) * 2**-40 ;  u2 = ((b & 0xffffff) + 0.5) * 2**-24
    r  = sqrt(-2 ln u1) ;  z0 = r cos(2 pi u2) ;  z1 = r sin(2 pi u2)

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


def uniform40(a, b):
    """Open-interval (0,1) uniform with 40 significant bits from the two words of a half block (exact in fp64)."""
    v = (a << np.uint64(8)) | (b >> np.uint64(24))
    return (v.astype(np.float64) + 0.5) * (2.0 ** -40)


def stream_words(seed, run, stream, j):
    """The two words (a, b) of one stream at sample(s) j: half (stream & 1) of block (j, stream >> 1)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    run = np.asarray(run, dtype=np.uint64)
    jj = np.asarray(j, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    w = philox4x32(jj, np.uint64(int(stream) >> 1), run & MASK32, run >> np.uint64(32), k0, k1)
    return (w[2], w[3]) if int(stream) & 1 else (w[0], w[1])


def box_muller(a, b):
    """(z0, z1) from the two words of a half block."""
    u1 = uniform40(a, b)
    u2 = ((b & np.uint64(0xFFFFFF)).astype(np.float64) + 0.5) * (2.0 ** -24)
    r = np.sqrt(-2.0 * np.log(u1))
    ang = (2.0 * np.pi) * u2
    return r * np.cos(ang), r * np.sin(ang)


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
