"""ctypes binding of libginsim.so (the C ABI declared in include/ginsim.h).

Same loading style as the reference's own FFI use (demo_algorithms/mag_calibrate.py:44,
demo_algorithms/aceinna_ins.py:172-173): ``cdll.LoadLibrary`` + caller-allocated NumPy buffers passed
as ``POINTER(c_double)``.  There is NO CPU fallback: if the library is missing, importing this module
raises; if no GPU is visible, creating a context raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GINSIM_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libginsim.so')   # GINSIM_LIB: A/B builds

ALGO_FREE = 1
ALGO_ODO = 2

OK, ERR_ARG, ERR_HIP, ERR_NODEV, ERR_RANGE, ERR_NOMEM, ERR_PLACED = 0, -1, -2, -3, -4, -5, -6


class GinsimError(RuntimeError):
    """HIP / device failure inside libginsim."""


class GinsimOutOfMemory(GinsimError):
    """The device is out of memory (GINSIM_ERR_NOMEM, ABI 7): the one failure worth a retry after giving memory back."""


class PlacedUnavailable(GinsimError):
    """This device has no usable placed arena (GINSIM_ERR_PLACED): allocate with ginsim_malloc instead."""


class PlacedOptions(C.Structure):
    """ginsim_placed_options (ABI 7); zero = the library's default."""
    _fields_ = [('stripe_bytes', C.c_int64), ('budget_bytes', C.c_int64), ('limit_bytes', C.c_int64), ('search_seconds', C.c_double)]


class PlacedInfo(C.Structure):
    """ginsim_placed_info (ABI 7)."""
    _fields_ = [('available', C.c_int32), ('classes', C.c_int32), ('searches', C.c_int32), ('failed', C.c_int32),
                ('stripe_bytes', C.c_int64), ('mapped_bytes', C.c_int64), ('used_bytes', C.c_int64), ('limit_bytes', C.c_int64),
                ('stripes_of_class', C.c_int64 * 3), ('chunks_created', C.c_int64), ('chunks_ambiguous', C.c_int64),
                ('probes', C.c_int64), ('peak_held_bytes', C.c_int64), ('search_seconds', C.c_double),
                ('last_search_seconds', C.c_double), ('anchor_ms', C.c_double), ('stripe_classes', C.c_char * 256)]


class SensorModel(C.Structure):
    _fields_ = [('bias', C.c_double * 3), ('gm_a', C.c_double * 3), ('gm_b', C.c_double * 3),
                ('white', C.c_double * 3), ('white_drift', C.c_int32 * 3), ('reserved', C.c_int32)]


VIB_PSD = 3


class Vibration(C.Structure):
    """ginsim_vibration (ABI 5): type 0 none / 1 random / 2 sinusoidal; ABI 8: 3 psd, a series on the device."""
    _fields_ = [('type', C.c_int32), ('random_phase', C.c_int32), ('amp', C.c_double * 3), ('omega_dt', C.c_double),
                ('series', C.c_void_p), ('period', C.c_int64)]


class McParams(C.Structure):
    _fields_ = [('n', C.c_int64), ('runs', C.c_int64), ('run_offset', C.c_uint64), ('seed', C.c_uint64),
                ('fs', C.c_double), ('ref_frame', C.c_int32), ('algo_mask', C.c_int32),
                ('earth_rot', C.c_int32), ('n_ini', C.c_int32), ('ini_first', C.c_uint64),
                ('ini_has_g', C.c_int32), ('given_sensors', C.c_int32),
                ('accel', SensorModel), ('gyro', SensorModel),
                ('odo_scale', C.c_double), ('odo_stdv', C.c_double), ('ref_end', C.c_double * 9),
                ('ini', C.c_void_p), ('ref_accel', C.c_void_p), ('ref_gyro', C.c_void_p), ('ref_odo', C.c_void_p),
                ('in_accel', C.c_void_p), ('in_gyro', C.c_void_p), ('in_odo', C.c_void_p),
                ('out_accel', C.c_void_p), ('out_gyro', C.c_void_p), ('out_odo', C.c_void_p),
                ('out_traj', C.c_void_p * 2), ('out_end', C.c_void_p * 2),
                ('wave_trace', C.c_void_p), ('block_threads', C.c_int32), ('end_pos_ned', C.c_int32),
                ('precision', C.c_int32), ('proc_pos_ned', C.c_int32),
                ('ref_nav', C.c_void_p), ('proc_first', C.c_int64), ('out_proc', C.c_void_p * 2),
                ('out_end_ned', C.c_void_p * 2), ('sensor_layout', C.c_int32), ('proc_plain_sums', C.c_int32),
                ('vib_accel', Vibration), ('vib_gyro', Vibration)]


class PathgenParams(C.Structure):
    _fields_ = [('ini_pva', C.c_double * 9), ('mobility', C.c_double * 3), ('fs', C.c_double),
                ('fs_gps', C.c_double), ('ref_frame', C.c_int32), ('enable_gps', C.c_int32),
                ('n_seg', C.c_int32), ('enable_mag', C.c_int32), ('geo_mag_n', C.c_double * 3)]


class AuxParams(C.Structure):
    _fields_ = [('n', C.c_int64), ('m', C.c_int64), ('runs', C.c_int64), ('run_offset', C.c_uint64),
                ('seed', C.c_uint64), ('gps_sigma', C.c_double * 6), ('mag_si', C.c_double * 9),
                ('mag_hi', C.c_double * 3), ('mag_std', C.c_double * 3), ('ref_gps', C.c_void_p),
                ('ref_mag', C.c_void_p), ('out_gps', C.c_void_p), ('out_mag', C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [('count', C.c_double), ('mean', C.c_double * 9), ('m2', C.c_double * 9),
                ('maxabs', C.c_double * 9)]


_PD = C.POINTER(C.c_double)
_SIGS = {
    'ginsim_abi_version': (C.c_int, []),
    'ginsim_last_error': (C.c_char_p, []),
    'ginsim_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'ginsim_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'ginsim_stream_first_xcc': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    'ginsim_destroy': (C.c_int, [C.c_void_p]),
    'ginsim_device_name': (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    'ginsim_mem_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    'ginsim_malloc': (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    'ginsim_free': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ginsim_placed_configure': (C.c_int, [C.c_void_p, C.POINTER(PlacedOptions)]),
    'ginsim_placed_reserve': (C.c_int, [C.c_void_p, C.c_size_t]),
    'ginsim_malloc_placed': (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    'ginsim_placed_release': (C.c_int, [C.c_void_p]),
    'ginsim_placed_info_get': (C.c_int, [C.c_void_p, C.POINTER(PlacedInfo)]),
    'ginsim_memcpy_h2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'ginsim_memcpy_d2h': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'ginsim_memset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    'ginsim_host_alloc': (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    'ginsim_host_free': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ginsim_sync': (C.c_int, [C.c_void_p]),
    'ginsim_timer_begin': (C.c_int, [C.c_void_p]),
    'ginsim_timer_end': (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    'ginsim_event_record': (C.c_int, [C.c_void_p, C.c_int32]),
    'ginsim_event_elapsed': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    'ginsim_pathgen_capacity': (C.c_int, [C.POINTER(PathgenParams), _PD, C.POINTER(C.c_int64)]),
    'ginsim_pathgen': (C.c_int, [C.POINTER(PathgenParams), _PD, C.c_int64, _PD, _PD, _PD, _PD, _PD,
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'ginsim_calc_true_sensor_output': (C.c_int, [_PD, _PD, _PD, _PD, _PD, _PD, C.c_int32, C.c_double, _PD, _PD, _PD, _PD]),
    'ginsim_parse_motion_def': (C.c_int, [_PD, _PD, _PD, _PD, _PD]),
    'ginsim_euler_update_zyx': (C.c_int, [_PD, _PD, C.c_double, _PD]),
    'ginsim_aux_sensors': (C.c_int, [C.c_void_p, C.POINTER(AuxParams)]),
    'ginsim_mc_run': (C.c_int, [C.c_void_p, C.POINTER(McParams)]),
    'ginsim_mc_variant': (C.c_int, [C.POINTER(McParams), C.POINTER(C.c_int32)]),
    'ginsim_mc_kernel_name': (C.c_int, [C.POINTER(McParams), C.c_char_p, C.c_size_t]),
    'ginsim_end_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(Stats)]),
    'ginsim_end_stats_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    'ginsim_end_stats_finish': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    'ginsim_comm_probe': (C.c_int, []),
    'ginsim_comm_unique_id': (C.c_int, [C.c_char_p]),
    'ginsim_comm_init': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]),
    'ginsim_comm_destroy': (C.c_int, [C.c_void_p]),
    'ginsim_comm_query': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'ginsim_vib_psd_series': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32,
                                        C.c_void_p]),
    'ginsim_end_stats_all_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    'ginsim_end_stats_all_finish': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    'ginsim_process_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _PD]),
    'ginsim_end_stats_from_traj': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                             C.POINTER(Stats)]),
    'ginsim_process_stats_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                           C.c_void_p, C.c_int32, C.c_uint64, _PD]),
    'ginsim_end_stats_from_traj_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                                 C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(Stats)]),
    'ginsim_stats_merge': (C.c_int, [C.POINTER(Stats), C.c_int32, C.POINTER(Stats)]),
    'ginsim_gather_runs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                     C.POINTER(C.c_int64), C.c_int32, _PD]),
    'ginsim_gather_series': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                       C.POINTER(C.c_int64), C.c_int32, _PD]),
    'ginsim_gather_runs_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                         C.POINTER(C.c_int64), C.c_int32, _PD]),
    'ginsim_free_integration': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, _PD, _PD, _PD,
                                          C.c_int64, C.c_int64, _PD, C.c_int32, C.c_int32, C.c_uint64,
                                          _PD, _PD, _PD]),
    'ginsim_allan': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_double, _PD, _PD,
                               C.POINTER(C.c_int32), C.c_int32]),
    'ginsim_runs_to_series': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    'ginsim_normal_transform': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64, _PD, _PD]),
    'ginsim_rng_normals': (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64, _PD, _PD,
                                     C.POINTER(C.c_uint32)]),
}
_OPTIONAL = {}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError('libginsim.so not built: run `python gnss-ins-sim_amd/build.py` '
                          '(or __graft_entry__.build()); expected %s' % LIB_PATH)
    lib = C.cdll.LoadLibrary(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in _OPTIONAL.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
EXPORTS = tuple(_SIGS)


def check(rc):
    """Translate a status code into the exception type the reference raises for the same mistake."""
    if rc == OK:
        return
    msg = lib.ginsim_last_error().decode('utf-8', 'replace')
    if rc in (ERR_ARG, ERR_RANGE):
        raise ValueError(msg)
    if rc == ERR_NOMEM:
        raise GinsimOutOfMemory(msg)
    if rc == ERR_PLACED:
        raise PlacedUnavailable(msg)
    raise GinsimError(msg)


def dptr(a):
    """POINTER(c_double) of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(_PD)
