"""ctypes access to the plain-C oracle (oracle/c/ginsim_oracle.c).  TEST ORACLE ONLY -- see
oracle/__init__.py for who may import this."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'c', 'ginsim_oracle.c')
LIB = os.path.join(HERE, '_build', 'libginsim_oracle.so')
_PD = C.POINTER(C.c_double)
_PF = C.POINTER(C.c_float)


TABLES = os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd', 'csrc')      # normal_tables.inc: committed constants of the stream


def build(force=False):
    newest = max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(TABLES, 'normal_tables.inc')))
    if force or not os.path.exists(LIB) or newest > os.path.getmtime(LIB):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-fopenmp', '-ffp-contract=off', '-std=c11', '-I' + TABLES,
                               '-o', LIB, SRC, '-lm'])
    return LIB


class SensorModel(C.Structure):
    _fields_ = [('bias', C.c_double * 3), ('gm_a', C.c_double * 3), ('gm_b', C.c_double * 3),
                ('white', C.c_double * 3), ('white_drift', C.c_int32 * 3), ('reserved', C.c_int32)]


class McParams(C.Structure):
    _fields_ = [('n', C.c_int64), ('runs', C.c_int64), ('run_offset', C.c_uint64), ('seed', C.c_uint64),
                ('fs', C.c_double), ('ref_frame', C.c_int32), ('algo_odo', C.c_int32), ('earth_rot', C.c_int32),
                ('n_ini', C.c_int32), ('ini_first', C.c_uint64), ('ini_has_g', C.c_int32), ('reserved', C.c_int32),
                ('accel', SensorModel), ('gyro', SensorModel), ('odo_scale', C.c_double), ('odo_stdv', C.c_double),
                ('ref_end', C.c_double * 9)]


class Vibration(C.Structure):
    _fields_ = [('type', C.c_int32), ('random_phase', C.c_int32), ('amp', C.c_double * 3), ('omega_dt', C.c_double)]


def _vibration(vib_def, fs, random_phase):
    """vib_def dict of Sim.__parse_env (ins_sim.py:642-701) -> vibration_t; None -> NULL."""
    if vib_def is None:
        return None
    v = Vibration()
    v.type = {'random': 1, 'sinusoidal': 2}[vib_def['type'].lower()]
    v.amp[:] = [float(vib_def['x']), float(vib_def['y']), float(vib_def['z'])]
    if v.type == 2:
        v.omega_dt = 2.0 * math.pi * float(vib_def['freq']) * (1.0 / fs)
        v.random_phase = int(random_phase)
    return v


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.cdll.LoadLibrary(build())
        _lib.oracle_mc_run.restype = C.c_int
        _lib.oracle_mc_run.argtypes = [C.POINTER(McParams), _PD, _PD, _PD, _PD, _PD, C.c_int64, _PD, _PD]
        _lib.oracle_mc_run_vib.restype = C.c_int
        _lib.oracle_mc_run_vib.argtypes = [C.POINTER(McParams), C.POINTER(Vibration), C.POINTER(Vibration), _PD, _PD, _PD, _PD, _PD,
                                           C.c_int64, _PD, _PD]
        _lib.oracle_free_integration.restype = None
        _lib.oracle_free_integration.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int64, _PD, _PD, _PD, _PD, C.c_int,
                                                 _PD, _PD, _PD]
        _lib.oracle_normals.restype = None
        _lib.oracle_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64, _PD, _PD]
        _lib.oracle_mc_run_f32.restype = C.c_int
        _lib.oracle_mc_run_f32.argtypes = [C.POINTER(McParams), _PD, _PD, _PD, _PD, _PD, C.c_int64, _PF, _PF, _PF]
        _lib.oracle_mc_run_f32_vib.restype = C.c_int
        _lib.oracle_mc_run_f32_vib.argtypes = [C.POINTER(McParams), C.POINTER(Vibration), C.POINTER(Vibration), _PD, _PD, _PD, _PD, _PD,
                                               C.c_int64, _PF, _PF, _PF]
        _lib.oracle_free_integration_f32_given.restype = None
        _lib.oracle_free_integration_f32_given.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int64, _PD, _PD, _PD, _PD, C.c_int,
                                                           _PF, _PF, _PF, _PD]
        _lib.oracle_allan_var.restype = C.c_int
        _lib.oracle_allan_var.argtypes = [_PD, C.c_int64, C.c_double, _PD, _PD]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_PD)


def _model(err, rw_key, fs):
    from .ins_np import gm_coeffs
    a, b, white = gm_coeffs(err['b_corr'], err['b_drift'], fs)
    m = SensorModel()
    for i in range(3):
        m.bias[i] = float(np.asarray(err['b'], dtype=np.float64)[i])
        m.gm_a[i], m.gm_b[i], m.white_drift[i] = float(a[i]), float(b[i]), int(white[i])
        m.white[i] = float(np.asarray(err[rw_key], dtype=np.float64)[i]) / np.sqrt(1.0 / fs)
    return m


def set_threads(k):
    os.environ['OMP_NUM_THREADS'] = str(int(k))


def mc_run(seed, run_offset, runs, fs, ref_frame, truth, accel_err, gyro_err, ini, algo='free', odo_err=None,
           earth_rot=True, ini_first=0, keep=0, vib_accel=None, vib_gyro=None):
    """Returns (end_err (runs,9), traj (keep,n,9) or None, sens (keep,n,6) or None).  vib_accel / vib_gyro: the reference's
    vib_def dicts ('random' | 'sinusoidal')."""
    ini = np.asarray(ini, dtype=np.float64)
    if ini.ndim == 1:
        ini = ini.reshape(-1, 1)
    table = np.zeros((ini.shape[1], 10))
    table[:, :min(10, ini.shape[0])] = ini[:10].T
    n = truth['ref_accel'].shape[0]
    p = McParams()
    p.n, p.runs, p.run_offset, p.seed, p.fs = n, int(runs), int(run_offset), int(seed) & (2 ** 64 - 1), float(fs)
    p.ref_frame, p.algo_odo, p.earth_rot = int(ref_frame), int(algo == 'odo'), int(bool(earth_rot))
    p.n_ini, p.ini_first, p.ini_has_g = table.shape[0], int(ini_first), int(ini.shape[0] > 9)
    p.accel, p.gyro = _model(accel_err, 'vrw', fs), _model(gyro_err, 'arw', fs)
    if odo_err is not None:
        p.odo_scale, p.odo_stdv = float(odo_err['scale']), float(odo_err['stdv'])
    end = np.concatenate([truth['ref_att'][-1], truth['ref_pos'][-1], truth['ref_vel'][-1]])
    p.ref_end[:] = [float(x) for x in end]
    ra = np.ascontiguousarray(truth['ref_accel'], dtype=np.float64)
    rg = np.ascontiguousarray(truth['ref_gyro'], dtype=np.float64)
    ro = np.ascontiguousarray(truth['ref_odo'], dtype=np.float64) if algo == 'odo' else None
    out = np.empty((int(runs), 9))
    traj = np.empty((keep, n, 9)) if keep else None
    sens = np.empty((keep, n, 6)) if keep else None
    va, vg = _vibration(vib_accel, fs, False), _vibration(vib_gyro, fs, True)
    rc = lib().oracle_mc_run_vib(C.byref(p), None if va is None else C.byref(va), None if vg is None else C.byref(vg),
                                 _p(table), _p(ra), _p(rg), _p(ro), _p(out), int(keep), _p(traj), _p(sens))
    if rc:
        raise MemoryError('oracle_mc_run')
    return out, traj, sens


def _pf(a):
    return None if a is None else a.ctypes.data_as(_PF)


def mc_run_f32(seed, run_offset, runs, fs, ref_frame, truth, accel_err, gyro_err, ini, algo='free', odo_err=None,
               earth_rot=True, ini_first=0, keep=0, vib_accel=None, vib_gyro=None):
    """The float restatement (oracle_mc_run_f32): what the fp32 kernel must reproduce to the bit.
    Returns (end_err (runs,9) float64, traj (keep,n,9) float32 = att3, position DISPLACEMENT3, vel3, sens (keep,n,6) float32,
    odo (keep,n) float32 or None)."""
    ini = np.asarray(ini, dtype=np.float64)
    if ini.ndim == 1:
        ini = ini.reshape(-1, 1)
    table = np.zeros((ini.shape[1], 10))
    table[:, :min(10, ini.shape[0])] = ini[:10].T
    n = truth['ref_accel'].shape[0]
    p = McParams()
    p.n, p.runs, p.run_offset, p.seed, p.fs = n, int(runs), int(run_offset), int(seed) & (2 ** 64 - 1), float(fs)
    p.ref_frame, p.algo_odo, p.earth_rot = int(ref_frame), int(algo == 'odo'), int(bool(earth_rot))
    p.n_ini, p.ini_first, p.ini_has_g = table.shape[0], int(ini_first), int(ini.shape[0] > 9)
    p.accel, p.gyro = _model(accel_err, 'vrw', fs), _model(gyro_err, 'arw', fs)
    if odo_err is not None:
        p.odo_scale, p.odo_stdv = float(odo_err['scale']), float(odo_err['stdv'])
    end = np.concatenate([truth['ref_att'][-1], truth['ref_pos'][-1], truth['ref_vel'][-1]])
    p.ref_end[:] = [float(x) for x in end]
    ra = np.ascontiguousarray(truth['ref_accel'], dtype=np.float64)
    rg = np.ascontiguousarray(truth['ref_gyro'], dtype=np.float64)
    ro = np.ascontiguousarray(truth['ref_odo'], dtype=np.float64) if (odo_err is not None and 'ref_odo' in truth) else None
    out = np.empty((int(runs), 9))
    traj = np.empty((keep, n, 9), dtype=np.float32) if keep else None
    sens = np.empty((keep, n, 6), dtype=np.float32) if keep else None
    odo = np.empty((keep, n), dtype=np.float32) if (keep and ro is not None) else None
    va, vg = _vibration(vib_accel, fs, False), _vibration(vib_gyro, fs, True)
    rc = lib().oracle_mc_run_f32_vib(C.byref(p), None if va is None else C.byref(va), None if vg is None else C.byref(vg),
                                     _p(table), _p(ra), _p(rg), _p(ro), _p(out), int(keep), _pf(traj), _pf(sens), _pf(odo))
    if rc:
        raise MemoryError('oracle_mc_run_f32')
    return out, traj, sens, odo


def free_integration_f32(ref_frame, fs, gyro, accel, ini, earth_rot=True, odo=None):
    """Given-data mechanisation in float (fp64 series rounded to float as read).  Returns att (n,3), position displacement
    (n,3), vel (n,3) as float32 and the fp64 end position (3,)."""
    g = np.ascontiguousarray(gyro, dtype=np.float64)
    a = None if accel is None else np.ascontiguousarray(accel, dtype=np.float64)
    o = None if odo is None else np.ascontiguousarray(odo, dtype=np.float64)
    ini = np.ascontiguousarray(np.asarray(ini, dtype=np.float64).reshape(-1))
    ini10 = np.zeros(10)
    ini10[:ini.size] = ini
    n = g.shape[0]
    att, dpos, vel = (np.empty((n, 3), dtype=np.float32) for _ in range(3))
    end_pos = np.empty(3)
    lib().oracle_free_integration_f32_given(int(ref_frame), float(fs), int(bool(earth_rot)), n, _p(g), _p(a), _p(o), _p(ini10),
                                            int(ini.size > 9), _pf(att), _pf(dpos), _pf(vel), _p(end_pos))
    return att, dpos, vel, end_pos


def free_integration(ref_frame, fs, gyro, accel, ini, earth_rot=True, odo=None):
    g = np.ascontiguousarray(gyro, dtype=np.float64)
    a = None if accel is None else np.ascontiguousarray(accel, dtype=np.float64)
    o = None if odo is None else np.ascontiguousarray(odo, dtype=np.float64)
    ini = np.ascontiguousarray(np.asarray(ini, dtype=np.float64).reshape(-1))
    ini10 = np.zeros(10)
    ini10[:ini.size] = ini
    n = g.shape[0]
    att, pos, vel = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 3))
    lib().oracle_free_integration(int(ref_frame), float(fs), int(bool(earth_rot)), n, _p(g), _p(a), _p(o), _p(ini10),
                                  int(ini.size > 9), _p(att), _p(pos), _p(vel))
    return att, pos, vel


def normals(seed, run, stream, count):
    z0, z1 = np.empty(count), np.empty(count)
    lib().oracle_normals(int(seed), int(run), int(stream), int(count), _p(z0), _p(z1))
    return z0, z1


def allan_var(x, fs):
    x = np.ascontiguousarray(x, dtype=np.float64)
    avar, tau = np.zeros(128), np.zeros(128)
    nt = lib().oracle_allan_var(_p(x), x.size, float(fs), _p(avar), _p(tau))
    return avar[:nt], tau[:nt]
