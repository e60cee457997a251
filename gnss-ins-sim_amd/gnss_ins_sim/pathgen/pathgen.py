r"""``gnss_ins_sim.pathgen.pathgen`` under its reference name (gnss_ins_sim/pathgen/pathgen.py): the truth generator
``path_gen`` and the sensor-error generators ``acc_gen`` / ``gyro_gen`` / ``gps_gen`` / ``odo_gen`` / ``mag_gen`` with the
reference's signatures, argument meaning, return values and exceptions -- over the C ABI of libginsim.so.

    path_gen   pathgen.py:26-329    -> ginsim_pathgen        (host C++ inside the library, one sequential recurrence)
    acc_gen    pathgen.py:441-501   \
    gyro_gen   pathgen.py:503-563    > ginsim_mc_run, algo_mask 0 (the sensor model of the fused kernel, one run, on the GPU)
    odo_gen    pathgen.py:627-641   /
    gps_gen    pathgen.py:596-625   \  ginsim_aux_sensors
    mag_gen    pathgen.py:643-661   /
    bias_drift               pathgen.py:565-594  -> the drift term of the same sensor model (ginsim_mc_run, everything else zero)
    calc_true_sensor_output  pathgen.py:331-411  -> ginsim_calc_true_sensor_output  (host C++, the function path_gen runs inline)
    parse_motion_def         pathgen.py:413-439  -> ginsim_parse_motion_def         (the same)

Differences from the reference, all stated: ``path_gen`` leaves ``motion_def`` and ``output_def`` unmodified (the reference
overwrites ``motion_def[:, 7]`` and ``output_def[1:, 1]``, pathgen.py:122, 137, 143); only ``simulation_over_sample_rate``
1 is built (every caller in the reference passes 1.0, ins_sim.py:459); the generators draw their noise from the engine's
counter-based stream with a key taken from ``np.random`` -- so ``np.random.seed`` makes them
repeatable, like the reference -- or from the keyword-only ``seed``; ``vib_def``: the 'random' and 'sinusoidal'
vibration models are carried by the kernels, a 'psd' one (time_series_from_psd.py) is outside the hot path.
There is no CPU implementation behind the generators: without a GPU they raise.
"""
import math

import numpy as np

VERSION = '1.0'
D2R = math.pi / 180


def path_gen(ini_pos_vel_att, motion_def, output_def, mobility, ref_frame=0, magnet=False, *, geo_mag_n=None,
             geo_mag_date=None):
    """pathgen.path_gen (pathgen.py:26-329).  Returns the reference's dict: 'status', 'imu' (n,7) = [idx, acc3, gyro3],
    'nav' (n,10) = [idx, pos3, velNED3, euler3], and -- when enabled, otherwise ``[]`` as in the reference -- 'mag' (n,4),
    'gps' (m,8) = [idx, pos3, vel3, visibility], 'odo' (n,5) = [idx, distance, vel_b3].

    magnet=True needs the geomagnetic field at the initial position (pathgen.py:164-168 evaluates the World Magnetic Model
    there): keyword ``geo_mag_n`` [uT, N frame], or a reachable checkout of the reference whose own geomag.py is then
    evaluated once on the host (geoparams.reference_geomag_n; ``geo_mag_date`` pins the date)."""
    import ginsim
    output_def = np.asarray(output_def, dtype=np.float64)
    motion_def = np.atleast_2d(np.asarray(motion_def, dtype=np.float64))
    ini = np.asarray(ini_pos_vel_att, dtype=np.float64)
    if output_def.shape[0] != 3 or output_def.ndim != 2:
        raise ValueError("output_def should be of size 3x2.")
    out_freq, sim_osr = output_def[0, 1], output_def[0, 0]
    if sim_osr != 1:
        raise NotImplementedError('path_gen: simulation_over_sample_rate %s; only 1 is built (every caller in the reference '
                                  'passes 1.0, ins_sim.py:459)' % sim_osr)
    total = 0
    for i in range(motion_def.shape[0]):
        if motion_def[i, 7] < 0:
            raise ValueError("Time duration of %s-th command has negative time duration: %s." % (i, motion_def[i, 7]))
        total += math.ceil(motion_def[i, 7] * out_freq)
    if total <= 0:
        raise ValueError("Total time duration in the motion definition file must be above 0.")
    enable_gps, enable_odo = output_def[1, 0] == 1, output_def[2, 0] == 1
    mag_n = None
    if magnet:
        mag_n = geo_mag_n
        if mag_n is None:
            from ..geoparams import geoparams
            mag_n = geoparams.reference_geomag_n(ini[0], ini[1], ini[2], geo_mag_date)
        if mag_n is None:
            raise NotImplementedError('path_gen(magnet=True) needs the local geomagnetic field: pass geo_mag_n=[bx, by, bz] uT or '
                                      'make a checkout of the reference reachable ($GNSS_INS_SIM_REFERENCE)')
    raw = ginsim.pathgen(ini, motion_def, out_freq, output_def[1, 1] if enable_gps else 0.0, mobility, ref_frame,
                         gps=bool(enable_gps), geo_mag_n=mag_n)
    # the engine hands out its remembered truth read-only; the reference's caller owns (and may overwrite) what it gets
    return {'status': True, 'imu': raw['imu'].copy(), 'nav': raw['nav'].copy(), 'mag': raw['mag'].copy() if magnet else [],
            'gps': raw['gps'].copy() if enable_gps else [], 'odo': raw['odo'].copy() if enable_odo else []}


def _key(seed):
    return int(np.random.randint(0, 2 ** 62)) if seed is None else int(seed)


def _one_run_sensors(fs, ref_accel, ref_gyro, acc_err, gyro_err, ref_odo=None, odo_err=None, seed=None, vib_accel=None,
                     vib_gyro=None):
    """One realisation of the IMU error model over given truth series, on the device (sensors-only launch)."""
    import ginsim
    n = ref_accel.shape[0]
    zeros = np.zeros((n, 3))
    truth = {'ref_accel': np.ascontiguousarray(ref_accel, dtype=np.float64), 'ref_gyro': np.ascontiguousarray(ref_gyro, dtype=np.float64),
             'ref_att': zeros, 'ref_pos': zeros, 'ref_vel': zeros}
    if ref_odo is not None:
        truth['ref_odo'] = np.ascontiguousarray(ref_odo, dtype=np.float64)
    job = ginsim.MonteCarloJob(ginsim.default_context(), fs, 0, truth, acc_err, gyro_err, None, runs=1, algos=(),
                               odo_err=odo_err, seed=_key(seed), keep_sensors=True, vib_accel=vib_accel, vib_gyro=vib_gyro).run()
    return job


_QUIET = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.array([np.inf, np.inf, np.inf]), 'arw': np.zeros(3),
          'vrw': np.zeros(3)}


def _psd_given_on_grid(vib_def, fs, n):
    """A 'psd' vib_def whose arrays lie on the series' own frequency grid is halved IN PLACE by the reference at every call
    (time_series_from_psd.py:44-49: no copy without the interpolation): leave the caller's arrays as the reference does."""
    import ginsim
    if vib_def is not None and str(vib_def['type']).lower() == 'psd':
        made = ginsim.psd_amplitudes(vib_def, fs, n)
        if made is not None and made[2]:
            for k in 'xyz':
                vib_def[k][1:-1] *= 0.5


def acc_gen(fs, ref_a, acc_err, vib_def=None, *, seed=None):
    """pathgen.acc_gen (pathgen.py:441-501): true specific force (n,3) + bias + Gauss-Markov drift + white noise + vibration
    (vib_def: {'type': 'random' | 'sinusoidal', 'x', 'y', 'z'[, 'freq']}, or {'type': 'psd', 'freq', 'x', 'y', 'z'} arrays)."""
    ref_a = np.asarray(ref_a, dtype=np.float64)
    job = _one_run_sensors(fs, ref_a, np.zeros_like(ref_a), acc_err, _QUIET, seed=seed, vib_accel=vib_def)
    out = job.sensors('accel', [0])[0]
    job.release()
    _psd_given_on_grid(vib_def, fs, ref_a.shape[0])
    return out


def gyro_gen(fs, ref_w, gyro_err, vib_def=None, *, seed=None):
    """pathgen.gyro_gen (pathgen.py:503-563); a sinusoidal vibration gets a uniform random phase per axis (:553-555)."""
    ref_w = np.asarray(ref_w, dtype=np.float64)
    job = _one_run_sensors(fs, np.zeros_like(ref_w), ref_w, _QUIET, gyro_err, seed=seed, vib_gyro=vib_def)
    out = job.sensors('gyro', [0])[0]
    job.release()
    _psd_given_on_grid(vib_def, fs, ref_w.shape[0])
    return out


def odo_gen(ref_odo, odo_err, *, seed=None):
    """pathgen.odo_gen (pathgen.py:627-641): scale * true forward speed + stdv * N."""
    ref_odo = np.asarray(ref_odo, dtype=np.float64).reshape(-1)
    z = np.zeros((ref_odo.shape[0], 3))
    job = _one_run_sensors(1.0, z, z, _QUIET, _QUIET, ref_odo=ref_odo, odo_err=odo_err, seed=seed)
    out = job.sensors('odo', [0])[0]
    job.release()
    return out


def gps_gen(ref_gps, gps_err, gps_type=0, *, seed=None):
    """pathgen.gps_gen (pathgen.py:596-625): gps_type 0 = LLA positions (stdp metres -> rad at the FIRST point), 1 = xyz."""
    import ginsim
    job = ginsim.AuxSensorJob(ginsim.default_context(), 1, seed=_key(seed), ref_gps=np.asarray(ref_gps, dtype=np.float64)[:, 0:6],
                              gps_err=gps_err, ref_frame=0 if gps_type == 0 else 1).run()
    out = job.series('gps', [0])[0]
    job.release()
    return out


def mag_gen(ref_mag, mag_err, *, seed=None):
    """pathgen.mag_gen (pathgen.py:643-661): si . (ref_mag + hi) + std * N."""
    import ginsim
    job = ginsim.AuxSensorJob(ginsim.default_context(), 1, seed=_key(seed), ref_mag=np.asarray(ref_mag, dtype=np.float64),
                              mag_err=mag_err).run()
    out = job.series('mag', [0])[0]
    job.release()
    return out


def bias_drift(corr_time, drift, n, fs, *, seed=None):
    """pathgen.bias_drift (pathgen.py:565-594): (n,3) bias-drift series -- first-order Gauss-Markov per axis with a finite
    correlation time, white N(0, drift) otherwise -- as the fused kernels generate it (sense3 in csrc/mc_kernel.hip): one
    sensors-only launch whose truth, constant bias and white noise are zero, so the measurement IS the drift."""
    err = {'b': np.zeros(3), 'b_drift': np.asarray(drift, dtype=np.float64) * np.ones(3),
           'b_corr': np.asarray(corr_time, dtype=np.float64) * np.ones(3), 'vrw': np.zeros(3)}
    zeros = np.zeros((int(n), 3))
    job = _one_run_sensors(fs, zeros, zeros, err, _QUIET, seed=seed)
    out = job.sensors('accel', [0])[0]
    job.release()
    return out


def calc_true_sensor_output(pos_n, vel_b, att, c_nb, vel_dot_b, att_dot, ref_frame, g):
    """pathgen.calc_true_sensor_output (pathgen.py:331-411): (acc, gyro, vel_dot_n, pos_dot_n) of one kinematic state; c_nb is
    the body -> nav matrix the caller holds for att (as in the reference), g is used in ref_frame 1 only."""
    from ginsim._lib import lib, check, dptr
    a = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1)) for v in (pos_n, vel_b, att, c_nb, vel_dot_b, att_dot)]
    if a[3].size != 9 or any(v.size != 3 for v in a[:3] + a[4:]):
        raise ValueError('calc_true_sensor_output: 3-vectors and a 3x3 matrix are expected')
    out = [np.empty(3) for _ in range(4)]
    check(lib.ginsim_calc_true_sensor_output(dptr(a[0]), dptr(a[1]), dptr(a[2]), dptr(a[3]), dptr(a[4]), dptr(a[5]),
                                             int(ref_frame), float(g), *[dptr(o) for o in out]))
    return tuple(out)


def parse_motion_def(motion_def_seg, att, vel):
    """pathgen.parse_motion_def (pathgen.py:413-439): (target attitude, target velocity) of a motion command of type 1..5."""
    from ginsim._lib import lib, check, dptr
    seg = np.ascontiguousarray(np.asarray(motion_def_seg, dtype=np.float64).reshape(-1))
    if seg.size < 7:
        raise ValueError('parse_motion_def: a segment has at least 7 elements')
    a = np.ascontiguousarray(np.asarray(att, dtype=np.float64).reshape(3))
    v = np.ascontiguousarray(np.asarray(vel, dtype=np.float64).reshape(3))
    att_com, vel_com = np.empty(3), np.empty(3)
    check(lib.ginsim_parse_motion_def(dptr(seg), dptr(a), dptr(v), dptr(att_com), dptr(vel_com)))
    return att_com, vel_com


def __getattr__(name):
    from .. import _reference
    return _reference.delegate(__name__, name)
