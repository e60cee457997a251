"""Counter-based normal generator of the MC engine, NumPy restatement (TEST ORACLE).

The reference draws its noise from the global legacy ``np.random.randn`` stream
(/root/reference/gnss_ins_sim/pathgen/pathgen.py:495,557,588,593,621-622,639,660)
which is serial and unseeded: "parity unpinned" for the stream itself.  The
engine therefore defines its own stream -- Philox4x32-10 (Salmon et al., SC'11,
"Parallel random numbers: as easy as 1, 2, 3"; Random123 v1.14 constants) followed
by Box-Muller in fp64 -- and parity is "identical injected normals": the
unmodified reference is fed THESE normals through a ``np.random.randn`` shim
(``oracle/ref_shim.py``).

Stream definition (shared by this file, ``oracle/c/ginsim_oracle.c`` and
``gnss-ins-sim_amd/csrc/philox.hpp``).  A normal pair needs a 53-bit uniform for the radius (tails to 8.5 sigma,
as NumPy's doubles) and a 32-bit uniform for the angle: 85 bits, so THREE pairs ("streams" 3g, 3g+1, 3g+2 of
group g) are cut from the 256 bits of TWO Philox blocks:

    key     = (seed & 0xffffffff, seed >> 32)
    A       = philox4x32_10((j, 2g,   run & 0xffffffff, run >> 32), key)      j = sample index
    B       = philox4x32_10((j, 2g+1, run & 0xffffffff, run >> 32), key)
    slot 0:  radius words (lo, hi) = (A0, A1)                                        angle word A2
    slot 1:  radius words (lo, hi) = (A3, B0)                                        angle word B1
    slot 2:  radius words (lo, hi) = ((A0 & 0x7ff) << 21 | (A3 & 0x7ff) << 10, B2)   angle word B3
             (a radius uses all of hi and the top 21 bits of lo: the low 11 bits of A0 and A3 are the spare ones)
    u1 = (((hi<<32 | lo) >> 11) + 0.5) * 2**-53 ;  u2 = (angle word + 0.5) * 2**-32
    r  = sqrt(-2 ln u1) ;  z0 = r cos(2 pi u2) ;  z1 = r sin(2 pi u2)

Stream ids (one stream -> two normals (z0, z1)):

    0: accel drift x, y     1: accel drift z, accel white x    2: accel white y, z        (group 0)
    3: gyro  drift x, y     4: gyro  drift z, gyro  white x    5: gyro  white y, z        (group 1)
    6: odometer, -          7: mag x, y                        8: mag z, -                (group 2)
    15: gps pos x, y       16: gps pos z, vel x               17: gps vel y, z            (group 5; j = GPS sample index)
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


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds.  All arguments broadcastable uint32-valued arrays.

    Returns four uint64 arrays holding 32-bit words.
    """
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for rnd in range(10):
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


def uniform53(lo, hi):
    """Open-interval (0,1) uniform from two 32-bit words (53 significant bits)."""
    v = ((hi << np.uint64(32)) | lo) >> np.uint64(11)
    return (v.astype(np.float64) + 0.5) * (2.0 ** -53)


def stream_words(seed, run, stream, j):
    """(lo, hi, angle) words of one stream: the radius uniform is uniform53(lo, hi), the angle uniform is
    (angle + 0.5) 2^-32.  See the module docstring for the cut of two Philox blocks into three streams."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    run = np.asarray(run, dtype=np.uint64)
    g, slot = divmod(int(stream), 3)
    jj = np.asarray(j, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    A = philox4x32_10(jj, np.uint64(2 * g), run & MASK32, run >> np.uint64(32), k0, k1)
    if slot == 0:
        return A[0], A[1], A[2]
    B = philox4x32_10(jj, np.uint64(2 * g + 1), run & MASK32, run >> np.uint64(32), k0, k1)
    if slot == 1:
        return A[3], B[0], B[1]
    lo = ((A[0] & np.uint64(0x7FF)) << np.uint64(21)) | ((A[3] & np.uint64(0x7FF)) << np.uint64(10))
    return lo, B[2], B[3]


def normal_pair(seed, run, stream, j):
    """Two standard normals (z0, z1) for (run, stream, sample j); arrays broadcast."""
    lo, hi, aw = stream_words(seed, run, stream, j)
    u1 = uniform53(lo, hi)
    u2 = (aw.astype(np.float64) + 0.5) * (2.0 ** -32)
    r = np.sqrt(-2.0 * np.log(u1))
    a = (2.0 * np.pi) * u2
    return r * np.cos(a), r * np.sin(a)


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
