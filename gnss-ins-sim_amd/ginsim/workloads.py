"""Named workloads of BASELINE.json (motion profile + IMU grade + frame), built through the C ABI.

Motion profiles under ../motion_profiles/ are the numeric content of the reference's
demo_motion_def_files/motion_def-{90deg_turn,long_drive,Allan}.csv, re-emitted by tests/golden/make_golden.py.
"""
import math
import os

import numpy as np

from . import engine

D2R = math.pi / 180.0
PROFILE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'motion_profiles')
HIGH_MOBILITY = (1.0, 0.5, 2.0)         # gnss_ins_sim/sim/ins_sim.py:25


def profile_path(name):
    return os.path.join(PROFILE_DIR, name + '.csv')


_PARSED = {}        # (path, mtime_ns, size) or the CSV text itself -> (ini, seg): np.genfromtxt twice costs 0.2 ms per Sim.run


def parse_motion(src):
    """Sim.__parse_motion (gnss_ins_sim/sim/ins_sim.py:578-610): file path or CSV text ->
    (ini_pva(9) [rad, m, m/s], motion_def(S,9) [rad]).  The parse of a file is remembered by (path, mtime, size), of a string by
    its text; callers get copies."""
    key = None
    try:
        if isinstance(src, str):
            if os.path.isfile(src):
                st = os.stat(src)
                key = (os.path.abspath(src), st.st_mtime_ns, st.st_size)
            else:
                key = src
            hit = _PARSED.get(key)
            if hit is not None:
                return hit[0].copy(), hit[1].copy()
    except OSError:
        key = None
    ini, seg = _parse_motion(src)
    if key is not None:
        if len(_PARSED) > 64:
            _PARSED.clear()
        _PARSED[key] = (ini.copy(), seg.copy())
    return ini, seg


def _parse_motion(src):
    try:
        if not os.path.isfile(src):
            src = list(src.split('\n'))
        ini = np.genfromtxt(src, delimiter=',', skip_header=1, max_rows=1)
        seg = np.genfromtxt(src, delimiter=',', skip_header=3)
        ini = np.array(ini[:9], dtype=np.float64)
        if seg.ndim == 1:
            seg = seg.reshape((1, len(seg)))
        seg = np.array(seg[:, :9], dtype=np.float64)
    except Exception:
        raise ValueError('motion definition file/string must have nine columns '
                         'and at least four rows (two header rows + at least two data rows).')
    ini[0] *= D2R
    ini[1] *= D2R
    ini[6:9] *= D2R
    seg[:, 1:4] *= D2R
    seg[np.isnan(seg)] = 0.0
    return ini, seg


def imu_grade(name):
    """(accel_err, gyro_err) of a built-in IMU grade in internal units.  Single source: the drop-in package's
    gnss_ins_sim.sim.imu_model (which restates gnss_ins_sim/sim/imu_model.py:18-52 and its unit conversions :138-143)."""
    from gnss_ins_sim.sim import imu_model
    imu = imu_model.IMU(accuracy=name, axis=6, gps=False)
    return imu.accel_err, imu.gyro_err


def truth_from_profile(name, fs, ref_frame, fs_gps=0.0, gps=False, mobility=HIGH_MOBILITY):
    """Run pathgen (C ABI) on a named profile; returns (ini_pva, truth dict, raw pathgen dict)."""
    ini, seg = parse_motion(profile_path(name))
    raw = engine.pathgen(ini, seg, fs, fs_gps, mobility, ref_frame, gps=gps)
    truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
             'ref_pos': np.ascontiguousarray(raw['nav'][:, 1:4]), 'ref_vel': np.ascontiguousarray(raw['nav'][:, 4:7]),
             'ref_att': np.ascontiguousarray(raw['nav'][:, 7:10]), 'ref_odo': np.ascontiguousarray(raw['odo'][:, 2]),
             'time': raw['nav'][:, 0] / fs}
    if gps:
        truth['ref_gps'] = np.ascontiguousarray(raw['gps'][:, 1:7])
        truth['gps_time'] = raw['gps'][:, 0] / fs
        truth['gps_visibility'] = raw['gps'][:, 7].copy()
    return ini, truth, raw
