"""NumPy restatement of the reference hot path (TEST ORACLE -- see oracle/__init__.py).

Vectorised over Monte-Carlo runs (leading axis R); the time loop stays a Python
loop because the recurrence is sequential.  All citations are relative to
/root/reference/.  Pinned against the unmodified reference by
tests/golden/make_golden.py (T1 given-data, T2 noise-free closed loop, T3 injected
noise) -- see tests/test_oracle_golden.py.
"""
import math
import numpy as np

from . import philox

# WGS-84 constants, gnss_ins_sim/geoparams/geoparams.py:17-23, 40-43
RE = 6378137.0
FLATTENING = 1 / 298.257223563
ECC = 0.0818191908426215
E_SQR = ECC ** 2
W_IE = 7292115e-11
G0 = 9.7803253359
GK = 0.00193185265241
GM_ = 0.00344978650684
TWO_PI = 2.0 * math.pi
HALF_PI = 0.5 * math.pi


# ----------------------------------------------------------------------------- earth model
def geo_param(lat, h):
    """geoparams.geo_param (geoparams.py:25-53), arrays in -> (rm, rn, g, sl, cl)."""
    sl = np.sin(lat)
    cl = np.cos(lat)
    s2 = sl * sl
    w = np.sqrt(1.0 - E_SQR * s2)
    rm = (RE * (1 - E_SQR)) / (w * (1.0 - E_SQR * s2))
    rn = RE / w
    g1 = G0 * (1 + GK * s2) / w
    g = g1 * (1.0 - (2.0 / RE) * (1.0 + FLATTENING + GM_ - 2.0 * FLATTENING * s2) * h
              + 3.0 * h * h / RE / RE)
    return rm, rn, g, sl, cl


def lla2ecef(lla):
    """geoparams.lla2ecef (geoparams.py:70-87); lla (...,3) -> (...,3)."""
    lla = np.asarray(lla, dtype=np.float64)
    sl, cl = np.sin(lla[..., 0]), np.cos(lla[..., 0])
    r = RE / np.sqrt(1.0 - E_SQR * sl * sl)
    rho = (r + lla[..., 2]) * cl
    return np.stack([rho * np.cos(lla[..., 1]), rho * np.sin(lla[..., 1]),
                     (r * (1.0 - E_SQR) + lla[..., 2]) * sl], axis=-1)


# ----------------------------------------------------------------------------- attitude leaves
def dcm_zyx(att):
    """attitude.euler2dcm(.,'zyx') (attitude.py:344-371): n->b DCM, att (...,3) -> (...,3,3)."""
    c, s = np.cos(att), np.sin(att)
    cy, cp, cr = c[..., 0], c[..., 1], c[..., 2]
    sy, sp, sr = s[..., 0], s[..., 1], s[..., 2]
    m = np.empty(att.shape[:-1] + (3, 3))
    m[..., 0, 0] = cp * cy
    m[..., 0, 1] = cp * sy
    m[..., 0, 2] = -sp
    m[..., 1, 0] = sr * sp * cy - cr * sy
    m[..., 1, 1] = sr * sp * sy + cr * cy
    m[..., 1, 2] = cp * sr
    m[..., 2, 0] = sp * cr * cy + sy * sr
    m[..., 2, 1] = sp * cr * sy - cy * sr
    m[..., 2, 2] = cp * cr
    return m


def euler_step_zyx(att, w, dt):
    """attitude.euler_update_zyx (attitude.py:679-721), (R,3) arrays."""
    cr, sr = np.cos(att[:, 2]), np.sin(att[:, 2])
    q = w[:, 2] * cr + w[:, 1] * sr
    yaw = att[:, 0] + q / np.cos(att[:, 1]) * dt
    pit = att[:, 1] + (w[:, 1] * cr - w[:, 2] * sr) * dt
    rol = att[:, 2] + (w[:, 0] + q * np.tan(att[:, 1])) * dt
    hi = pit > HALF_PI
    lo = (~hi) & (pit < -HALF_PI)
    fold = hi | lo
    pit = np.where(hi, math.pi - pit, np.where(lo, -math.pi - pit, pit))
    yaw = np.where(fold, yaw + math.pi, yaw)
    rol = np.where(fold, rol + math.pi, rol)
    yaw = np.where(yaw > math.pi, yaw - TWO_PI, np.where(yaw < -math.pi, yaw + TWO_PI, yaw))
    rol = np.where(rol > math.pi, rol - TWO_PI, np.where(rol < -math.pi, rol + TWO_PI, rol))
    return np.stack([yaw, pit, rol], axis=1)


def cross(a, b):
    """attitude.cross3 (attitude.py:758-770) on (...,3)."""
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def angle_range_pi(x):
    """attitude.angle_range_pi (attitude.py:799-812)."""
    x = np.mod(x, TWO_PI)
    return np.where(x > math.pi, x - TWO_PI, x)


def euler_range_three_axis(a):
    """attitude.euler_angle_range_three_axis (attitude.py:772-797) for one 3-vector."""
    a1, a2, a3 = a[0], float(angle_range_pi(a[1])), a[2]
    if a2 > HALF_PI:
        a2, a1, a3 = math.pi - a2, a1 + math.pi, a3 + math.pi
    elif a2 < -HALF_PI:
        a2, a1, a3 = -math.pi - a2, a1 + math.pi, a3 + math.pi
    return np.array([float(angle_range_pi(a1)), a2, float(angle_range_pi(a3))])


# ----------------------------------------------------------------------------- truth (pathgen)
def path_gen(ini_pva, motion_def, fs, fs_gps, mobility, ref_frame, gps=False, odo=False, geo_mag_n=None):
    """pathgen.path_gen (pathgen.py:26-329) with sim_osr == 1 (ins_sim.py:451).

    ``motion_def`` is (S,9) with angles already in rad (ins_sim.py:604-608); it is NOT
    modified (the reference overwrites column 7 in place, pathgen.py:122).
    Returns dict imu(n,7) nav(n,10) [gps(m,8)] [odo(n,5)].
    """
    motion_def = np.array(motion_def, dtype=np.float64, copy=True)
    dt = 1.0 / fs
    alpha = 0.9
    max_acc, max_dw, max_w = mobility
    kp, kd = 5.0, 10.0
    seg_steps = np.empty(motion_def.shape[0])
    total = 0
    for i in range(motion_def.shape[0]):
        if motion_def[i, 7] < 0:
            raise ValueError("Time duration of %s-th command has negative time duration: %s."
                             % (i, motion_def[i, 7]))
        c = motion_def[i, 7] * fs
        total += math.ceil(c)
        seg_steps[i] = round(c)
    if total <= 0:
        raise ValueError("Total time duration in the motion definition file must be above 0.")
    total = int(total)
    imu = np.zeros((total, 7))
    nav = np.zeros((total, 10))
    gps_every = round(fs / fs_gps) if gps else 0
    gps_rows = np.zeros((total, 8))
    odo_rows = np.zeros((total, 5))
    mag_rows = np.zeros((total, 4))
    if geo_mag_n is not None:                       # pathgen.py:164-171
        geo_mag_n = np.array(geo_mag_n, dtype=np.float64)
        if ref_frame == 1:
            geo_mag_n[0] = math.sqrt(geo_mag_n[0] * geo_mag_n[0] + geo_mag_n[1] * geo_mag_n[1])
            geo_mag_n[1] = 0.0

    pos0 = np.array(ini_pva[0:3], dtype=np.float64)
    vel_b = np.array(ini_pva[3:6], dtype=np.float64)
    att = np.array(ini_pva[6:9], dtype=np.float64)
    c_nb = dcm_zyx(att).T
    vel_n = c_nb.dot(vel_b)
    dpos = np.zeros(3)
    g0 = float(geo_param(pos0[0], pos0[2])[2])
    if ref_frame == 1:
        pos0 = lla2ecef(pos0)
    att_dot = np.zeros(3)
    vel_dot_b = np.zeros(3)
    k = 0
    kg = 0
    odo_dist = 0.0
    for i in range(motion_def.shape[0]):
        typ = round(motion_def[i, 0])
        vis = motion_def[i, 8]
        cmd_a = motion_def[i, 1:4]
        cmd_v = motion_def[i, 4:7]
        # pathgen.parse_motion_def (pathgen.py:413-439); compares the raw value to 1..5
        raw = motion_def[i, 0]
        if raw == 1 or raw == 2:
            tgt_a, tgt_v = cmd_a.copy(), cmd_v.copy()
        elif raw == 3:
            tgt_a, tgt_v = att + cmd_a, vel_b + cmd_v
        elif raw == 4:
            tgt_a, tgt_v = cmd_a.copy(), vel_b + cmd_v
        elif raw == 5:
            tgt_a, tgt_v = att + cmd_a, cmd_v.copy()
        else:
            raise ValueError('unsupported motion type %s' % raw)
        filt_a, filt_v = att, vel_b
        stop = k + seg_steps[i]
        done = False
        while k < stop and not done:
            if typ == 1:
                att_dot = alpha * att_dot + (1 - alpha) * tgt_a
                vel_dot_b = alpha * vel_dot_b + (1 - alpha) * tgt_v
            else:
                filt_a = alpha * filt_a + (1 - alpha) * tgt_a
                filt_v = alpha * filt_v + (1 - alpha) * tgt_v
                vel_dot_b = np.clip((filt_v - vel_b) / dt, -max_acc, max_acc)
                add = np.clip(kp * (tgt_a - att) + kd * (0 - att_dot), -max_dw, max_dw)
                att_dot = np.clip(att_dot + add * dt, -max_w, max_w)
                da, dv = att - tgt_a, vel_b - tgt_v
                if math.sqrt(da.dot(da)) < 1e-4 and math.sqrt(dv.dot(dv)) < 1e-4:
                    done = True
            acc, gyro, pos_dot = true_sensor_output(pos0 + dpos, vel_b, att, c_nb, vel_dot_b,
                                                    att_dot, ref_frame, g0)
            imu[k, 0] = k
            imu[k, 1:4] = acc
            imu[k, 4:7] = gyro
            nav[k, 0] = k
            nav[k, 1:4] = pos0 + dpos
            nav[k, 4:7] = vel_n
            nav[k, 7:10] = euler_range_three_axis(att)
            odo_rows[k] = (k, odo_dist, vel_b[0], vel_b[1], vel_b[2])
            if geo_mag_n is not None:               # pathgen.py:273-279
                mag_rows[k, 0] = k
                mag_rows[k, 1:4] = c_nb.T.dot(geo_mag_n)
            if gps and (k % gps_every) == 0:
                gps_rows[kg, 0] = k
                gps_rows[kg, 1:4] = pos0 + dpos
                gps_rows[kg, 4:7] = vel_n
                gps_rows[kg, 7] = vis
                kg += 1
            dpos = dpos + pos_dot * dt
            odo_dist = odo_dist + math.sqrt(vel_b.dot(vel_b)) * dt
            vel_b = vel_b + vel_dot_b * dt
            att = att + att_dot * dt
            c_nb = dcm_zyx(att).T
            vel_n = c_nb.dot(vel_b)
            k += 1
        if done:
            att_dot = np.zeros(3)
            vel_dot_b = np.zeros(3)
    out = {'imu': imu[:k], 'nav': nav[:k]}
    if gps:
        out['gps'] = gps_rows[:kg]
    if odo:
        out['odo'] = odo_rows[:k]
    if geo_mag_n is not None:
        out['mag'] = mag_rows[:k]
    return out


def true_sensor_output(pos_n, vel_b, att, c_nb, vel_dot_b, att_dot, ref_frame, g):
    """pathgen.calc_true_sensor_output (pathgen.py:331-411) -> (acc, gyro, pos_dot_n)."""
    vel_n = c_nb.dot(vel_b)
    w_en_n = np.zeros(3)
    w_ie_n = np.zeros(3)
    if ref_frame == 0:
        rm, rn, g, sl, cl = (float(v) for v in geo_param(pos_n[0], pos_n[2]))
        rm_e, rn_e = rm + pos_n[2], rn + pos_n[2]
        w_en_n[0] = vel_n[1] / rn_e
        w_en_n[1] = -vel_n[0] / rm_e
        w_en_n[2] = -vel_n[1] * sl / cl / rn_e
        w_ie_n[0] = W_IE * cl
        w_ie_n[2] = -W_IE * sl
        pos_dot = np.array([vel_n[0] / rm_e, vel_n[1] / rn_e / cl, -vel_n[2]])
    else:
        pos_dot = vel_n.copy()
    sh, ch = math.sin(att[0]), math.cos(att[0])
    w_nb_n = np.array([-sh * att_dot[1] + c_nb[0, 0] * att_dot[2],
                       ch * att_dot[1] + c_nb[1, 0] * att_dot[2],
                       att_dot[0] + c_nb[2, 0] * att_dot[2]])
    gyro = c_nb.T.dot(w_nb_n + w_en_n + w_ie_n)
    w_ie_b = c_nb.T.dot(w_ie_n)
    acc = vel_dot_b + cross(w_ie_b + gyro, vel_b) - c_nb.T.dot(np.array([0.0, 0.0, g]))
    return acc, gyro, pos_dot


# ----------------------------------------------------------------------------- sensor errors
def gm_coeffs(corr, drift, fs):
    """AR(1) coefficients of pathgen.bias_drift (pathgen.py:583-586); inf corr -> a=0, b=drift."""
    corr = np.asarray(corr, dtype=np.float64)
    drift = np.asarray(drift, dtype=np.float64)
    a = np.zeros(3)
    b = np.zeros(3)
    white = np.isinf(corr)
    for i in range(3):
        if white[i]:
            b[i] = drift[i]
        else:
            a[i] = 1 - 1 / fs / corr[i]
            b[i] = drift[i] * np.sqrt(1.0 - np.exp(-2 / (fs * corr[i])))
    return a, b, white


def vibration_series(fs, n, vib_def, nv=None, u=None):
    """The vib term of pathgen.acc_gen / gyro_gen (pathgen.py:476-492 / 538-556) for R runs: (R,n,3) (or (1,n,3) when it is
    the same for every run).  vib_def: the dict Sim.__parse_env makes ('random' | 'sinusoidal'; 'psd': psd_series);
    nv (R,n,3): the normals of a 'random' vibration; u (R,3) or None: the uniforms of a sinusoidal vibration's random phases
    (gyro_gen: np.random.rand(1) per axis; None = phase 0, acc_gen)."""
    amp = np.array([vib_def['x'], vib_def['y'], vib_def['z']], dtype=np.float64)
    kind = vib_def['type'].lower()
    if kind == 'random':
        return amp * nv                                                     # pathgen.py:486-488
    if kind == 'sinusoidal':
        dt = 1.0 / fs
        arg = 2.0 * math.pi * vib_def['freq'] * dt * np.arange(n)           # pathgen.py:490, left to right
        if u is None:
            return (amp * np.sin(arg)[:, None])[None]
        phase = np.asarray(u, dtype=np.float64) * 2 * math.pi               # np.random.rand(1)*2*math.pi (:553)
        return amp * np.sin(arg[None, :, None] + phase[:, None, :])
    raise NotImplementedError(kind)


def psd_series(vib_def, fs, n, z, calls_before=0):
    """time_series_from_psd.time_series_from_psd (time_series_from_psd.py:16-63) for the three axes of one run, as pathgen.acc_gen /
    gyro_gen call it (pathgen.py:479-484 / :541-546): (n, 3), or zeros where the reference returns status False.
    z (L, 3): the normals np.random.randn(L) returned for x, y, z (:52).  calls_before: how many times the reference has already been
    through this PSD -- it halves a PSD GIVEN on the series' own grid in place at every call (:44-49: no copy without the
    interpolation), so run r of a batch starts from 0.5^r of the array; nothing is mutated here."""
    freq = np.asarray(vib_def['freq'], dtype=np.float64)
    out = np.zeros((n, 3))
    if fs < 2.0 * freq[-1] or fs < 0.0:
        return out
    N = n + (n % 2)
    repeat = N != n
    if N > 16384:
        N, repeat = 16384, True
    L = N // 2 + 1
    for c, key in enumerate('xyz'):
        sxx = np.array(vib_def[key], dtype=np.float64)
        if freq.shape[0] != L:
            sxx = np.interp(np.linspace(0, fs / 2.0, L), freq, sxx)
        else:
            sxx[1:L - 1] = sxx[1:L - 1] * 0.5 ** calls_before
        sxx[1:L - 1] = 0.5 * sxx[1:L - 1]
        ax = np.sqrt(sxx * N * fs)
        xk = ax * np.exp(1j * (math.pi * z[:, c]))
        xk = np.hstack([xk, xk[-2:0:-1].conj()])
        x = np.fft.ifft(xk).real
        out[:, c] = np.hstack([np.tile(x, (n // N,)), x[0:n % N]]) if repeat else x
    return out


def psd_len(n):
    """L of time_series_from_psd for a series of n samples."""
    return min(n + (n % 2), 16384) // 2 + 1


def sensor_errors(fs, ref, err, rw_key, nd, nw, vib=None):
    """pathgen.acc_gen / gyro_gen (pathgen.py:441-501 / 503-563); vib: the (R|1,n,3) vibration term (vibration_series) or None.

    ref (n,3) truth; err dict with 'b','b_drift','b_corr', rw_key in {'vrw','arw'};
    nd, nw (R,n,3) drift / white normals.  Returns (R,n,3).
    """
    dt = 1.0 / fs
    R, n, _ = nw.shape
    a, b, white = gm_coeffs(err['b_corr'], err['b_drift'], fs)
    d = np.zeros((R, n, 3))
    for i in range(3):
        if white[i]:
            d[:, :, i] = b[i] * nd[:, :, i]             # pathgen.py:593
        else:
            for j in range(1, n):                        # pathgen.py:589-590
                d[:, j, i] = a[i] * d[:, j - 1, i] + b[i] * nd[:, j - 1, i]
    noise = nw * (np.asarray(err[rw_key], dtype=np.float64) / math.sqrt(dt))
    out = ref[None, :, :] + np.asarray(err['b'], dtype=np.float64) + d + noise
    return out if vib is None else out + vib                                # pathgen.py:500, 562: the vibration is added last



def odo_errors(ref_odo, odo_err, nz):
    """pathgen.odo_gen (pathgen.py:627-641); ref_odo (n,), nz (R,n)."""
    return odo_err['scale'] * ref_odo[None, :] + odo_err['stdv'] * nz


def gps_errors(ref_gps, gps_err, ref_frame, npos, nvel):
    """pathgen.gps_gen (pathgen.py:596-625); ref_gps (m,6); npos,nvel (R,m,3)."""
    pos_err = np.array(gps_err['stdp'], dtype=np.float64)
    if ref_frame == 0:
        rm, rn, _, _, cl = geo_param(ref_gps[0, 0], ref_gps[0, 2])
        pos_err[0] = pos_err[0] / rm
        pos_err[1] = pos_err[1] / rn / cl
    return np.concatenate([ref_gps[None, :, 0:3] + pos_err * npos,
                           ref_gps[None, :, 3:6] + np.asarray(gps_err['stdv']) * nvel], axis=2)


def mag_errors(ref_mag, mag_err, nz):
    """pathgen.mag_gen (pathgen.py:643-661); ref_mag (n,3), nz (R,n,3)."""
    m = (ref_mag + mag_err['hi']).dot(np.asarray(mag_err['si']).T)
    return m[None] + np.asarray(mag_err['std']) * nz


# ----------------------------------------------------------------------------- mechanisation
def free_integration(ref_frame, fs, gyro, accel, ini, earth_rot=True, odo=None):
    """demo_algorithms/free_integration.py:63-174 (odo is None) or
    free_integration_odo.py:63-160 (odo (R,n) given), batched over runs.

    gyro, accel: (R,n,3) (accel ignored when odo is given); ini: (9|10,) or (9|10,R).
    Returns att, pos, vel each (R,n,3).
    """
    R, n, _ = gyro.shape
    dt = 1.0 / fs
    ini = np.asarray(ini, dtype=np.float64)
    if ini.ndim == 1:
        ini = np.repeat(ini[:, None], R, axis=1)
    r0, v0, a0 = ini[0:3].T.copy(), ini[3:6].T.copy(), ini[6:9].T.copy()
    g_ext = ini[9].copy() if ini.shape[0] > 9 else None
    att = np.zeros((R, n, 3))
    pos = np.zeros((R, n, 3))
    vel = np.zeros((R, n, 3))
    att[:, 0] = a0
    vel_b = v0.copy()
    C = dcm_zyx(att[:, 0])                                     # n -> b
    vel[:, 0] = np.einsum('rji,rj->ri', C, vel_b)              # C^T vel_b
    if ref_frame == 1:
        pos[:, 0] = lla2ecef(r0)
        g = geo_param(r0[:, 0], r0[:, 2])[2] if g_ext is None else g_ext
        for i in range(1, n):
            w = gyro[:, i - 1]
            att[:, i] = euler_step_zyx(att[:, i - 1], w, dt)
            if odo is None:
                cg = C[:, :, 2] * g[:, None]                   # C . [0,0,g]
                vel_b = vel_b + (accel[:, i - 1] + cg) * dt - cross(w, vel_b) * dt
            else:
                vel_b = np.zeros((R, 3))
                vel_b[:, 0] = odo[:, i - 1]
            C = dcm_zyx(att[:, i])
            vel[:, i] = np.einsum('rji,rj->ri', C, vel_b)
            pos[:, i] = pos[:, i - 1] + vel[:, i - 1] * dt
    else:
        pos[:, 0] = r0
        for i in range(1, n):
            p, v = pos[:, i - 1], vel[:, i - 1]
            rm, rn, g, sl, cl = geo_param(p[:, 0], p[:, 2])
            rm_e, rn_e = rm + p[:, 2], rn + p[:, 2]
            if g_ext is not None:
                g = g_ext
            w_en = np.stack([v[:, 1] / rn_e, -v[:, 0] / rm_e, -v[:, 1] * sl / cl / rn_e], axis=1)
            w_ie = np.zeros((R, 3))
            if earth_rot:
                w_ie[:, 0] = W_IE * cl
                w_ie[:, 2] = -W_IE * sl
            w_nb_b = gyro[:, i - 1] - np.einsum('rij,rj->ri', C, w_en + w_ie)
            att[:, i] = euler_step_zyx(att[:, i - 1], w_nb_b, dt)
            if odo is None:
                gn = np.zeros((R, 3))
                gn[:, 2] = g
                vdot = np.einsum('rji,rj->ri', C, accel[:, i - 1]) + gn - cross(2 * w_ie + w_en, v)
                vel[:, i] = v + vdot * dt
            pos[:, i, 0] = p[:, 0] + v[:, 0] / rm_e * dt
            pos[:, i, 1] = p[:, 1] + v[:, 1] / rn_e / cl * dt
            pos[:, i, 2] = p[:, 2] + (-v[:, 2]) * dt
            C = dcm_zyx(att[:, i])
            if odo is not None:
                vb = np.zeros((R, 3))
                vb[:, 0] = odo[:, i - 1]
                vel[:, i] = np.einsum('rji,rj->ri', C, vb)
    return att, pos, vel


# ----------------------------------------------------------------------------- end-to-end MC
def imu_err_dicts(imu):
    """Plain-dict copy of the four error dicts of an imu_model.IMU-like object."""
    return ({k: np.array(v, dtype=np.float64) for k, v in imu.accel_err.items()},
            {k: np.array(v, dtype=np.float64) for k, v in imu.gyro_err.items()})


def mc_vibration(seed, runs, fs, n, vib_def, sensor):
    """The vibration term of MC runs ``runs`` with the engine's counter RNG; sensor 'acc' | 'gyr' (gyro_gen draws random
    phases for a sinusoidal vibration, acc_gen does not)."""
    if vib_def is None:
        return None
    kind = vib_def['type'].lower()
    if kind == 'psd':       # the phases of bin k are the normals a 'random' vibration would draw at sample k; run r is call r
        return np.stack([psd_series(vib_def, fs, n, philox.vib_normals(seed, r, psd_len(n), sensor), calls_before=int(r)) for r in runs])
    if kind == 'random':
        return vibration_series(fs, n, vib_def, nv=np.stack([philox.vib_normals(seed, r, n, sensor) for r in runs]))
    u = np.stack([philox.vib_phase_uniforms(seed, r, sensor) for r in runs]) if sensor == 'gyr' else None
    return vibration_series(fs, n, vib_def, u=u)


def mc_sensors(seed, runs, fs, ref_accel, ref_gyro, accel_err, gyro_err, vib_accel=None, vib_gyro=None):
    """Sensor data of MC runs ``runs`` (1-D int array of global run ids) with the engine's
    counter RNG, following the loop body of Sim.__gen_data_from_pathgen (ins_sim.py:490-506)."""
    n = ref_accel.shape[0]
    z = [philox.imu_normals(seed, r, n) for r in runs]
    nd_a = np.stack([x['acc_d'] for x in z])
    nw_a = np.stack([x['acc_w'] for x in z])
    nd_g = np.stack([x['gyr_d'] for x in z])
    nw_g = np.stack([x['gyr_w'] for x in z])
    accel = sensor_errors(fs, ref_accel, accel_err, 'vrw', nd_a, nw_a, mc_vibration(seed, runs, fs, n, vib_accel, 'acc'))
    gyro = sensor_errors(fs, ref_gyro, gyro_err, 'arw', nd_g, nw_g, mc_vibration(seed, runs, fs, n, vib_gyro, 'gyr'))
    return accel, gyro


def mc_odo(seed, runs, ref_odo, odo_err):
    n = ref_odo.shape[0]
    return odo_errors(ref_odo, odo_err, np.stack([philox.odo_normals(seed, r, n) for r in runs]))


# ----------------------------------------------------------------------------- statistics
def end_point_errors(att, pos, vel, ref_att, ref_pos, ref_vel):
    """Last-sample error of each run: InsDataMgr.array_error + __end_point_error_stats
    (ins_data_manager.py:519-541, 717-759).  Returns (R,9) = [att3 (wrapped), pos3, vel3]."""
    ea = angle_range_pi(att[:, -1] - ref_att[-1])
    return np.concatenate([ea, pos[:, -1] - ref_pos[-1], vel[:, -1] - ref_vel[-1]], axis=1)


def lla_error_ned(x, r):
    """array_error(lla=1) (ins_data_manager.py:542-552): LLA error in metres in the NED frame at r; (...,3)."""
    d = lla2ecef(x) - lla2ecef(r)
    sl, cl, so, co = np.sin(r[..., 0]), np.cos(r[..., 0]), np.sin(r[..., 1]), np.cos(r[..., 1])
    return np.stack([-sl * co * d[..., 0] - sl * so * d[..., 1] + cl * d[..., 2],
                     -so * d[..., 0] + co * d[..., 1],
                     -cl * co * d[..., 0] - cl * so * d[..., 1] - sl * d[..., 2]], axis=-1)


def process_error_stats(att, pos, vel, ref_att, ref_pos, ref_vel, first_sample, pos_ned=False):
    """InsDataMgr.__process_error_stats (ins_data_manager.py:761-795) for every run: (R,3,9) = max|e|, mean, std."""
    ea = angle_range_pi(att - ref_att[None])
    ep = lla_error_ned(pos, np.broadcast_to(ref_pos[None], pos.shape)) if pos_ned else pos - ref_pos[None]
    e = np.concatenate([ea, ep, vel - ref_vel[None]], axis=2)[:, first_sample:]
    return np.stack([np.max(np.abs(e), 1), np.mean(e, 1), np.std(e, 1)], axis=1)


def array_stats(e):
    """InsDataMgr.__array_stats (ins_data_manager.py:797-808)."""
    return {'max': np.max(np.abs(e), 0), 'avg': np.average(e, 0), 'std': np.std(e, 0)}


# ----------------------------------------------------------------------------- Allan variance
def allan_var(x, fs):
    """allan.allan_var (gnss_ins_sim/allan/allan.py:18-59)."""
    ts = 1.0 / fs
    n = len(x)
    mmax = int(math.floor(n / 9.0))
    if mmax * ts < 1:
        return np.zeros(0), np.zeros(0)
    mult = []
    scale = 0.1
    for _ in range(math.ceil(math.log10(mmax))):
        scale *= 10
        for j in range(1, 10):
            m = int(j * scale)
            if m > mmax:
                break
            mult.append(m)
    avar = np.zeros(len(mult))
    tau = np.zeros(len(mult))
    for i, m in enumerate(mult):
        nb = n // m
        if nb < 9:
            break
        means = x[:nb * m].reshape(nb, m).mean(axis=1)
        d = np.diff(means)
        avar[i] = 0.5 / (nb - 1) * np.sum(d * d)
        tau[i] = m * ts
    return avar, tau
