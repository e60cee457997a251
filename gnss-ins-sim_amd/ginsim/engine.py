"""Host-side driver objects over the C ABI: device context/buffers, truth generation, and one
Monte-Carlo batch (``MonteCarloJob``) = one launch of the fused HIP kernel on one GPU.

Nothing here computes on the CPU: every number a job returns was produced by libginsim on the device
(pathgen excepted, which is native host code inside the same library and runs once per Sim.run).
"""
import ctypes as C
import os
import math

import numpy as np

from . import _lib
from ._lib import lib, check, dptr, ALGO_FREE, ALGO_ODO

ALGO_BITS = {'free': ALGO_FREE, 'odo': ALGO_ODO}
ALGO_SLOT = {'free': 0, 'odo': 1}


def device_count():
    n = C.c_int(0)
    rc = lib.ginsim_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class DeviceBuffer(object):
    """A hipMalloc'ed region owned by a Context.  Large regions go back to the context's pool when they are freed and are
    handed out again for the next request of the same size: a hipMalloc of the 3-5 GB series buffers of a 65 536-run Sim
    takes anything between 1 and 280 ms on MI355X (measured: 226 / 3 / 124 / 284 ms in four consecutive Sim.run calls), the
    kernel that fills them 1.3 ms.  Everything of a context runs on its one stream, so a region that is reused is only
    touched after the work that last used it."""

    def __init__(self, ctx, nbytes, placed=False):
        """placed=True: carve the region from the device's placed arena (ABI 7: a range whose 512 MiB stripes cycle through the
        three classes of physical memory of an MI355X, so that a launch streaming several planes at once writes 6.8-7.0 instead of
        5.7-5.9 TB/s); where the device has no arena the region is hipMalloc'ed as before and ``placed`` stays False."""
        self.ctx, self.nbytes, self.placed = ctx, int(nbytes), False
        self.ptr = None
        if placed and ctx.placed_enabled:
            p = C.c_void_p()
            rc = lib.ginsim_malloc_placed(ctx.handle, self.nbytes, C.byref(p))
            if rc == 0:
                self.ptr, self.placed = p.value, True
                return
            ctx._placed_refused(rc)          # raises for anything but "no arena here" / out of memory
        self.ptr = ctx._pool_take(self.nbytes)
        if self.ptr is None:
            p = C.c_void_p()
            rc = lib.ginsim_malloc(ctx.handle, self.nbytes, C.byref(p))
            if rc != 0 and ctx._pool_bytes:          # out of memory with regions parked in the pool: give them back and retry
                ctx.release_pool()
                rc = lib.ginsim_malloc(ctx.handle, self.nbytes, C.byref(p))
            check(rc)
            self.ptr = p.value

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)

    def free(self, pool=True):
        """pool=False: hipFree now, whatever the pool would take."""
        if self.ptr and self.ctx.handle:
            if self.placed:                 # back to the arena's free list: no driver call, nothing to pool
                lib.ginsim_free(self.ctx.handle, self.ptr)
            elif not (pool and self.ctx._pool_give(self.nbytes, self.ptr)):
                lib.ginsim_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView(object):
    """A sub-range of a DeviceBuffer (not owning)."""

    def __init__(self, parent, byte_offset, nbytes, layout='runs'):
        self.parent, self.ptr, self.nbytes = parent, parent.ptr + int(byte_offset), int(nbytes)
        self.layout = layout        # 'runs': [axis][sample][run] (run fastest); 'series': [run][axis][sample]

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)

    def free(self):
        self.ptr = None


class Context(object):
    """One GPU (one HIP stream).  Raises GinsimError when no GPU is visible -- there is no CPU path."""

    _created = 0            # contexts of this process so far: `serial` orders them (= the order their streams were created in)

    def __init__(self, device=0):
        h = C.c_void_p()
        self.handle = None
        check(lib.ginsim_create(int(device), C.byref(h)))
        self.handle = h.value
        self.device = int(device)
        Context._created += 1
        self.serial = Context._created
        self.comm_ranks = 0
        # freed device regions by size (DeviceBuffer): regions of at least POOL_MIN bytes, pool_limit bytes in total -- at most a
        # third of the device's memory (96 GiB of an MI355X's 288 GB), so that on a smaller GPU the pool cannot sit on most of
        # the HBM; allocations the LIBRARY makes (scratch regions, gather buffers) and other users of the device are served by
        # draining the pool when they run out of memory (retry_oom)
        self._pool, self._pool_bytes = {}, 0
        self._comm_abandoned = None         # set when a communicator bootstrap timed out on a helper thread (distributed.py)
        limit = os.environ.get('GINSIM_POOL_BYTES')
        if limit is None:
            total = self.mem_info()[1]
            limit = min(96 * 2 ** 30, total // 3) if total else 96 * 2 ** 30
        self.pool_limit = int(limit)
        # placed memory (ABI 7): on unless $GINSIM_PLACED == '0'; switched off for this context when the device turns out to
        # have no usable arena (the reason is kept in placed_note)
        self.placed_enabled = os.environ.get('GINSIM_PLACED', '1') != '0'
        self.placed_note = None if self.placed_enabled else '$GINSIM_PLACED == 0'
        self._placed_configured = False

    # ---- placed memory
    PLACED_MIN_JOB = 1 << 30        # a job whose materialised series reach this many bytes carves them from the placed arena

    def _placed_configure(self):
        """$GINSIM_PLACED_STRIPE_MIB / _BUDGET_GIB / _LIMIT_GIB / _SEARCH_S -> ginsim_placed_configure, once and before the first
        placed request of this context (an arena another context of the device built already keeps its own options)."""
        if self._placed_configured:
            return
        self._placed_configured = True
        o, any_set = _lib.PlacedOptions(), False
        for env, field, scale in (('GINSIM_PLACED_STRIPE_MIB', 'stripe_bytes', 1 << 20), ('GINSIM_PLACED_BUDGET_GIB', 'budget_bytes', 1 << 30),
                                  ('GINSIM_PLACED_LIMIT_GIB', 'limit_bytes', 1 << 30), ('GINSIM_PLACED_SEARCH_S', 'search_seconds', None)):
            v = os.environ.get(env)
            if v:
                setattr(o, field, float(v) if scale is None else int(float(v) * scale))
                any_set = True
        if any_set and lib.ginsim_placed_configure(self.handle, C.byref(o)) != 0:
            self.placed_note = 'options ignored: ' + lib.ginsim_last_error().decode('utf-8', 'replace')

    def _placed_refused(self, rc):
        """A placed request failed: "no arena on this device" and "out of memory" mean hipMalloc from here on, anything else raises."""
        if rc not in (_lib.ERR_PLACED, _lib.ERR_NOMEM):
            check(rc)
        self.placed_note = lib.ginsim_last_error().decode('utf-8', 'replace')
        if rc == _lib.ERR_PLACED:
            self.placed_enabled = False

    def placed_reserve(self, nbytes):
        """Grow the device's arena so that `nbytes` more can be carved from it: ONE search for all regions of a job.  False (and
        placed memory off for this context) where the device has no usable arena."""
        if not self.placed_enabled:
            return False
        self._placed_configure()
        # Growing means a search (0.3 - 4 s).  Jobs that are no longer reachable but sit in reference cycles -- a Sim the script has
        # replaced by the next one -- still hold their regions until the collector runs: let it run first when what is free would
        # not do (a script that makes one Sim after another then re-uses the same stripes: 3 ms instead of a search per Sim)
        i = _lib.PlacedInfo()
        if lib.ginsim_placed_info_get(self.handle, C.byref(i)) == 0 and i.mapped_bytes > 0 and i.mapped_bytes - i.used_bytes < int(nbytes):
            import gc
            gc.collect()
        rc = lib.ginsim_placed_reserve(self.handle, int(nbytes))
        if rc != 0:
            self._placed_refused(rc)
            return False
        return True

    def first_xcc(self):
        """The XCD on which the first workgroup of a launch on this context's stream lands (ginsim_stream_first_xcc)."""
        v = C.c_int32(-1)
        check(lib.ginsim_stream_first_xcc(self.handle, C.byref(v)))
        return int(v.value)

    def placed_info(self):
        """ginsim_placed_info of the device's arena as a dict (sizes in bytes, times in seconds)."""
        i = _lib.PlacedInfo()
        check(lib.ginsim_placed_info_get(self.handle, C.byref(i)))
        out = {k: getattr(i, k) for k, _ in _lib.PlacedInfo._fields_ if k not in ('stripes_of_class', 'stripe_classes')}
        out['stripes_of_class'] = list(i.stripes_of_class)
        out['stripe_classes'] = i.stripe_classes.decode('ascii', 'replace')
        out['enabled'] = bool(self.placed_enabled)
        if self.placed_note:
            out['note'] = self.placed_note
        return out

    def mem_info(self):
        """(free, total) bytes of the device (hipMemGetInfo); (0, 0) if the query fails."""
        f, t = C.c_size_t(0), C.c_size_t(0)
        if lib.ginsim_mem_info(self.handle, C.byref(f), C.byref(t)) != 0:
            return 0, 0
        return int(f.value), int(t.value)

    def retry_oom(self, call):
        """call() -> status of a library entry point.  When it fails for lack of device memory while regions are parked in the
        pool, the pool is given back to the driver and the call is made once more (the library's own allocations -- scratch
        regions, gather buffers -- do not know about the pool)."""
        rc = call()
        if rc == _lib.ERR_NOMEM:                            # hipErrorOutOfMemory only (ABI 7): no other failure is retried
            import gc
            gc.collect()                                    # unreachable jobs in reference cycles park their regions in the pool now
            if self._pool_bytes:
                self.release_pool()
                rc = call()
        return rc

    def name(self):
        buf = C.create_string_buffer(256)
        check(lib.ginsim_device_name(self.handle, buf, 256))
        return buf.value.decode()

    POOL_MIN = 1 << 20

    def malloc(self, nbytes, placed=False):
        if placed:
            self._placed_configure()
        return DeviceBuffer(self, nbytes, placed=placed)

    def _pool_take(self, nbytes):
        lst = self._pool.get(nbytes)
        if not lst:
            return None
        self._pool_bytes -= nbytes
        return lst.pop()

    def _pool_give(self, nbytes, ptr):
        if nbytes < self.POOL_MIN or self._pool_bytes + nbytes > self.pool_limit:
            return False
        self._pool.setdefault(nbytes, []).append(ptr)
        self._pool_bytes += nbytes
        return True

    def release_pool(self):
        """hipFree everything parked in the pool and give the placed arena back if nothing is carved from it (also done by close())."""
        for lst in self._pool.values():
            for ptr in lst:
                lib.ginsim_free(self.handle, ptr)
        self._pool, self._pool_bytes = {}, 0
        if self.handle:
            lib.ginsim_placed_release(self.handle)

    def upload(self, array):
        a = np.ascontiguousarray(array)
        buf = DeviceBuffer(self, max(a.nbytes, 8))
        check(lib.ginsim_memcpy_h2d(self.handle, buf.ptr, a.ctypes.data, a.nbytes))
        return buf

    def download(self, buf_or_ptr, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        ptr = getattr(buf_or_ptr, 'ptr', buf_or_ptr)          # DeviceBuffer, DeviceView or a raw device pointer
        check(lib.ginsim_memcpy_d2h(self.handle, out.ctypes.data, ptr, out.nbytes))
        return out

    def sync(self):
        check(lib.ginsim_sync(self.handle))

    def timer_begin(self):
        check(lib.ginsim_timer_begin(self.handle))

    def timer_end(self):
        ms = C.c_float(0)
        check(lib.ginsim_timer_end(self.handle, C.byref(ms)))
        return ms.value

    def event_record(self, slot):
        check(lib.ginsim_event_record(self.handle, int(slot)))

    def event_elapsed(self, slot_a, slot_b):
        ms = C.c_float(0)
        check(lib.ginsim_event_elapsed(self.handle, int(slot_a), int(slot_b), C.byref(ms)))
        return ms.value

    # ---- multi-GPU exchange behind the C ABI (RCCL on this context's stream)
    @staticmethod
    def comm_probe():
        """Raises unless librccl can be reached from this process (dlopen + dlsym only)."""
        check(lib.ginsim_comm_probe())

    @staticmethod
    def comm_unique_id():
        """128 bytes rank 0 hands to the other ranks (ncclGetUniqueId)."""
        buf = C.create_string_buffer(128)
        check(lib.ginsim_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        check(lib.ginsim_comm_init(self.handle, int(nranks), int(rank), bytes(unique_id)))
        self.comm_ranks = int(nranks)

    def comm_query(self):
        """(ranks, rank, device) as the communicator itself reports them (ncclCommCount / UserRank / CuDevice; -1 = query missing)."""
        n, r, d = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        check(lib.ginsim_comm_query(self.handle, C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value

    def comm_destroy(self):
        if self.handle:
            check(lib.ginsim_comm_destroy(self.handle))
        self.comm_ranks = 0

    def close(self):
        if self.handle:
            t = self._comm_abandoned
            if t is not None and t.is_alive():
                return          # ncclCommInitRank is still running on this handle (distributed.init_abi_comm timed out): leak it
            self.release_pool()
            lib.ginsim_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = None


def default_context():
    """Process-wide context on device LOCAL_RANK (or 0); created on first use."""
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None or _DEFAULT_CTX.handle is None:
        import os
        _DEFAULT_CTX = Context(int(os.environ.get('LOCAL_RANK', '0')))
    return _DEFAULT_CTX


# --------------------------------------------------------------------------------------- truth
_TRUTH = {}          # arguments -> result of the last few truth generations (deterministic; long_drive @200 Hz takes 0.2 s)


def pathgen(ini_pva, motion_def, fs, fs_gps=0.0, mobility=(1.0, 0.5, 2.0), ref_frame=0, gps=False, geo_mag_n=None):
    """pathgen.path_gen through the C ABI.  motion_def is (S,9) with angles in rad, unmodified.
    geo_mag_n: geomagnetic field [uT] in the N frame at the initial position -> also emits 'mag' (n,4).
    Returns {'imu': (n,7), 'nav': (n,10), 'odo': (n,5)[, 'gps': (m,8)][, 'mag': (n,4)]}.  The truth is a pure function of
    the arguments: the last results are remembered and handed out again as READ-ONLY arrays."""
    md = np.ascontiguousarray(np.atleast_2d(np.asarray(motion_def, dtype=np.float64)))
    if md.shape[1] < 9:
        raise ValueError('motion definition must have nine columns')
    md = np.ascontiguousarray(md[:, :9])
    key = (np.asarray(ini_pva, dtype=np.float64)[:9].tobytes(), md.tobytes(), float(fs), float(fs_gps),
           tuple(float(x) for x in mobility), int(ref_frame), bool(gps),
           None if geo_mag_n is None else tuple(float(x) for x in geo_mag_n))
    hit = _TRUTH.get(key)
    if hit is not None:
        return dict(hit)
    out = _pathgen(ini_pva, md, fs, fs_gps, mobility, ref_frame, gps, geo_mag_n)
    for a in out.values():
        a.setflags(write=False)
    if len(_TRUTH) >= 4:
        _TRUTH.pop(next(iter(_TRUTH)))
    _TRUTH[key] = out
    return dict(out)


def _pathgen(ini_pva, md, fs, fs_gps, mobility, ref_frame, gps, geo_mag_n):
    p = _lib.PathgenParams()
    p.ini_pva[:] = [float(x) for x in np.asarray(ini_pva, dtype=np.float64)[:9]]
    p.mobility[:] = [float(x) for x in mobility]
    p.fs, p.fs_gps = float(fs), float(fs_gps)
    p.ref_frame, p.enable_gps, p.n_seg = int(ref_frame), int(bool(gps)), md.shape[0]
    if geo_mag_n is not None:
        p.enable_mag = 1
        p.geo_mag_n[:] = [float(x) for x in geo_mag_n]
    cap = C.c_int64(0)
    check(lib.ginsim_pathgen_capacity(C.byref(p), dptr(md), C.byref(cap)))
    cap = cap.value
    imu = np.zeros((cap, 7))
    nav = np.zeros((cap, 10))
    odo = np.zeros((cap, 5))
    gpsb = np.zeros((cap, 8)) if gps else None
    magb = np.zeros((cap, 4)) if geo_mag_n is not None else None
    n, m = C.c_int64(0), C.c_int64(0)
    check(lib.ginsim_pathgen(C.byref(p), dptr(md), cap, dptr(imu), dptr(nav), dptr(gpsb), dptr(odo), dptr(magb),
                             C.byref(n), C.byref(m)))
    out = {'imu': imu[:n.value], 'nav': nav[:n.value], 'odo': odo[:n.value]}
    if gps:
        out['gps'] = gpsb[:m.value]
    if magb is not None:
        out['mag'] = magb[:n.value]
    return out


# --------------------------------------------------------------------------------------- sensor models
def sensor_model(err, rw_key, fs):
    """Coefficients of pathgen.bias_drift / acc_gen / gyro_gen (pathgen.py:583-593, 496, 558) from an
    imu_model error dict ('b', 'b_drift', 'b_corr', rw_key)."""
    m = _lib.SensorModel()
    b = np.asarray(err['b'], dtype=np.float64) * np.ones(3)
    drift = np.asarray(err['b_drift'], dtype=np.float64) * np.ones(3)
    corr = np.asarray(err['b_corr'], dtype=np.float64) * np.ones(3)
    rw = np.asarray(err[rw_key], dtype=np.float64) * np.ones(3)
    dt = 1.0 / fs
    for i in range(3):
        m.bias[i] = b[i]
        m.white[i] = rw[i] / math.sqrt(dt)
        if math.isinf(corr[i]):
            m.gm_a[i], m.gm_b[i], m.white_drift[i] = 0.0, drift[i], 1
        else:
            m.gm_a[i] = 1 - 1 / fs / corr[i]
            m.gm_b[i] = drift[i] * math.sqrt(1.0 - math.exp(-2 / (fs * corr[i])))
            m.white_drift[i] = 0
    return m


VIB_TYPES = {'random': 1, 'sinusoidal': 2}


def vibration(vib_def, fs, random_phase):
    """ginsim_vibration from the reference's vib_def dict ({'type': 'random' | 'sinusoidal', 'x', 'y', 'z'[, 'freq']}, what
    Sim.__parse_env makes of an env string, ins_sim.py:642-701).  random_phase: gyro_gen draws one uniform phase per run and
    axis for a sinusoidal vibration (pathgen.py:553-555), acc_gen uses phase 0 (:490-492).  A 'psd' definition (an (n, 4) env
    array, ins_sim.py:686-697) needs its series on the device first: MonteCarloJob makes them (psd_amplitudes,
    ginsim_vib_psd_series) and fills type 3 in itself."""
    v = _lib.Vibration()
    if vib_def is None:
        return v
    kind = str(vib_def['type']).lower()
    if kind == 'psd':
        raise ValueError("a 'psd' vibration is a series per run: MonteCarloJob(vib_accel=..., vib_gyro=...) makes it on the device")
    if kind not in VIB_TYPES:
        raise ValueError('unknown vibration type %r' % (vib_def['type'],))
    v.type = VIB_TYPES[kind]
    v.amp[:] = [float(vib_def['x']), float(vib_def['y']), float(vib_def['z'])]
    if kind == 'sinusoidal':
        dt = 1.0 / fs
        v.omega_dt = 2.0 * math.pi * float(vib_def['freq']) * dt           # the reference's product, left to right
        v.random_phase = int(bool(random_phase))
    return v


PSD_MAX_PERIOD = 16384          # time_series_from_psd.py:41-43


def psd_amplitudes(vib_def, fs, n):
    """The host part of time_series_from_psd.time_series_from_psd (time_series_from_psd.py:16-50) for the three axes of one
    sensor: (period N, amplitudes (3, N // 2 + 1), given_on_grid) -- or None where the reference returns zeros (:32-34: the PSD
    reaches beyond fs / 2).
      N = n, n + 1 when n is odd, at most 16384 (:36-43; the series is tiled to n, :58-63)
      the PSD interpolated to linspace(0, fs / 2, N // 2 + 1) unless it is GIVEN on that many points (:44-48)
      interior bins halved (single- to double-sided, :49), a = sqrt(sxx N fs) (:50)
    given_on_grid: the reference then halves the CALLER'S array in place at every call (no copy was made): run r of a batch
    sees 0.5^(r + 1).  The amplitudes returned are then those of the array as it is now, nothing halved; the device applies the
    run's factor (ginsim_vib_psd_series, halve_per_run) and the caller of a Sim mutates the arrays afterwards as the reference does."""
    freq = np.asarray(vib_def['freq'], dtype=np.float64)
    fs = float(fs)
    if fs < 2.0 * freq[-1] or fs < 0.0:
        return None
    N = int(n) + (int(n) % 2)
    N = min(N, PSD_MAX_PERIOD)
    L = N // 2 + 1
    on_grid = freq.shape[0] == L
    amp = np.empty((3, L))
    for c, k in enumerate('xyz'):
        sxx = np.asarray(vib_def[k], dtype=np.float64)
        if not on_grid:
            sxx = np.interp(np.linspace(0, fs / 2.0, L), freq, sxx)
            sxx[1:L - 1] = 0.5 * sxx[1:L - 1]
        amp[c] = np.sqrt(sxx * N * fs)
    return N, amp, on_grid


def ini_table(ini):
    """(9|10,) or (9|10,k) initial states (free_integration.py:47-61) -> ((k,10) table, has_g)."""
    a = np.asarray(ini, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    elif a.ndim != 2:
        raise ValueError('Initial states should be a 1D or 2D numpy array, but the dimension is %s.' % a.ndim)
    if a.shape[0] < 9:
        raise ValueError('initial states need at least 9 elements')
    has_g = a.shape[0] > 9
    t = np.zeros((a.shape[1], 10))
    t[:, :min(a.shape[0], 10)] = a[:10].T
    return t, has_g


class StatsResult(object):
    """count / mean / M2 / max|e| of the 9 end-point error components (att3, pos3, vel3)."""

    def __init__(self, s):
        self.count = float(s.count)
        self.mean = np.array(s.mean[:])
        self.m2 = np.array(s.m2[:])
        self.maxabs = np.array(s.maxabs[:])

    @staticmethod
    def zero():
        """The record of a rank that holds no runs (neutral element of the merge)."""
        return StatsResult(_lib.Stats())

    @property
    def std(self):      # np.std(ddof=0), ins_data_manager.py:808
        return np.sqrt(self.m2 / self.count)

    def pack(self):
        return np.concatenate([[self.count], self.mean, self.m2, self.maxabs])

    @staticmethod
    def unpack(v):
        s = _lib.Stats()
        s.count = v[0]
        s.mean[:] = list(v[1:10])
        s.m2[:] = list(v[10:19])
        s.maxabs[:] = list(v[19:28])
        return s

    @staticmethod
    def merge(packed_rows):
        """Merge packed partials (one row per device) with the library's Chan merge."""
        rows = [r for r in packed_rows if r[0] > 0]
        if not rows:
            return StatsResult.zero()
        arr = (_lib.Stats * len(rows))(*[StatsResult.unpack(r) for r in rows])
        out = _lib.Stats()
        check(lib.ginsim_stats_merge(arr, len(rows), C.byref(out)))
        return StatsResult(out)


def starts_on_truth(table, nav0, ref_frame=0):
    """Do all the initial states of `table` ([n_ini][10]: pos3 LLA, body velocity3, yaw / pitch / roll, g) lie ON the truth's first
    sample `nav0` (att3, pos3, vel3)?  Then the error at sample 0 is zero for every run, the launch-wide shift of the online process
    statistics would be nine zeros, and the launch may take its sums as they are (ginsim_mc_params.proc_plain_sums: the same
    numbers, nine subtractions per step fewer).  Only ref_frame 0 has the form (there the truth's position is LLA like the
    table's).  The position must agree exactly, the attitude to 1e-12 rad modulo a turn, the velocity -- the body velocity turned
    into the navigation frame here, pathgen's own there -- to 1e-9 m/s: the plain sums' floor under such means is 1e-8 of them."""
    if int(ref_frame) != 0:
        return False
    nav0 = np.asarray(nav0, dtype=np.float64)
    for row in np.asarray(table, dtype=np.float64).reshape(-1, 10):
        datt = np.mod(row[6:9] - nav0[0:3] + np.pi, 2.0 * np.pi) - np.pi
        if not (np.all(np.abs(datt) <= 1e-12) and np.array_equal(row[0:3], nav0[3:6])):
            return False
        c, s = np.cos(row[6:9]), np.sin(row[6:9])
        n2b = np.array([[c[1] * c[0], c[1] * s[0], -s[1]],
                        [s[2] * s[1] * c[0] - c[2] * s[0], s[2] * s[1] * s[0] + c[2] * c[0], c[1] * s[2]],
                        [s[1] * c[2] * c[0] + s[0] * s[2], s[1] * c[2] * s[0] - c[0] * s[2], c[1] * c[2]]])
        if not np.all(np.abs(n2b.T @ row[3:6] - nav0[6:9]) <= 1e-9):
            return False
    return True


class MonteCarloJob(object):
    """One batch of MC runs on one device: fused noise injection + mechanisation + end-point error.

    truth: dict with 'ref_accel' (n,3), 'ref_gyro' (n,3), 'ref_att'/'ref_pos'/'ref_vel' (n,3) and, for
    the odometer algorithm, 'ref_odo' (n,).  algos: subset of ('free', 'odo').

    proc_first: None, or the first sample of the process-error window: max|e| / mean / std of the error over the samples
    >= proc_first are accumulated online inside the kernel (InsDataMgr.__process_error_stats, ins_data_manager.py:761-795)
    -- the statistics Sim.results() prints by default, without keeping a single trajectory.  One algorithm per job.
    proc_ned / end_ned (ref_frame 0): position errors of those statistics / of a second end-point record in local NED
    metres (extra_opt='ned', ins_data_manager.py:542-552).

    given: None (sensors are generated), or a dict of DeviceBuffers {'gyro', 'accel'[, 'odo']} holding sensor series
    that are already on the device in the engine's [component][sample][run] fp64 layout -- e.g. the 'gyro'/'accel'
    buffers another job materialised -- which are then integrated as they are (the plugin's run(set_of_input)
    boundary for a whole batch; accel_err / gyro_err may be None).  precision='f32' rounds them to float as it reads them.

    Sensors-only jobs (algos=()) of few runs and long series (<= 1024 runs, >= 2048 samples) run on the time-parallel series
    kernels and keep their series SERIES-major, [run][axis][sample] (``sensor_layout == 'series'``; ginsim_mc_params.sensor_layout
    1): that is the layout ginsim_allan reads, so ``allan()`` needs no re-layout.  ``sensors()`` hides the difference.

    vib_accel / vib_gyro: the reference's vib_def dicts ({'type': 'random' | 'sinusoidal', 'x', 'y', 'z'[, 'freq']},
    ins_sim.py:642-701) -> the vibration term of pathgen.acc_gen / gyro_gen (pathgen.py:476-492, 538-556).

    placed: None (default) -- the materialised series are carved from the device's placed arena (Context.placed_*, ABI 7) when
    they reach Context.PLACED_MIN_JOB bytes together; True / False force it.  ``placement()`` says what happened.
    """

    def __init__(self, ctx, fs, ref_frame, truth, accel_err, gyro_err, ini, runs, algos=('free',),
                 odo_err=None, earth_rot=True, seed=0, run_offset=0, ini_first=0,
                 keep_sensors=False, keep_traj=False, end_pos_ned=False, precision='f64', given=None,
                 proc_first=None, proc_ned=False, end_ned=False, vib_accel=None, vib_gyro=None, placed=None):
        self.ctx = ctx
        self.algos = tuple(algos)
        for a in self.algos:
            if a not in ALGO_BITS:
                raise ValueError('unknown algorithm %r' % (a,))
        self.n = int(truth['ref_accel'].shape[0])
        self.runs = int(runs)
        self.keep_sensors, self.keep_traj = bool(keep_sensors), bool(keep_traj)
        self.want_odo = 'odo' in self.algos or (odo_err is not None and 'ref_odo' in truth)
        if given is not None:
            if keep_sensors or not self.algos:
                raise ValueError('given sensors: at least one algorithm, nothing to keep but trajectories')
            need = ['gyro'] + (['accel'] if 'free' in self.algos else []) + (['odo'] if 'odo' in self.algos else [])
            for k in need:
                size = (1 if k == 'odo' else 3) * self.n * self.runs * 8
                if k not in given or given[k].nbytes < size:
                    raise ValueError('given sensors: %r missing or smaller than %d bytes' % (k, size))
                if getattr(given[k], 'layout', 'runs') != 'runs':
                    raise ValueError('given sensors: %r is series-major ([run][axis][sample]); the mechanisation reads '
                                     '[axis][sample][run]' % (k,))
            self.want_odo = False
        p = self.params = _lib.McParams()
        p.n, p.runs, p.run_offset, p.seed = self.n, self.runs, int(run_offset), int(seed) & (2 ** 64 - 1)
        p.fs, p.ref_frame = float(fs), int(ref_frame)
        p.algo_mask = sum(ALGO_BITS[a] for a in self.algos)
        p.earth_rot = int(bool(earth_rot))
        p.end_pos_ned = int(bool(end_pos_ned))
        if precision not in ('f64', 'f32'):
            raise ValueError("precision must be 'f64' or 'f32'")
        self.precision = precision
        p.precision = 1 if precision == 'f32' else 0
        self._esize = 4 if precision == 'f32' else 8
        if ini is None:
            if self.algos:
                raise ValueError('initial states are required when an algorithm is integrated')
            table, has_g = np.zeros((1, 10)), False
        else:
            table, has_g = ini_table(ini)
        p.n_ini, p.ini_first, p.ini_has_g, p.given_sensors = table.shape[0], int(ini_first), int(has_g), 0
        if given is None:
            p.accel = sensor_model(accel_err, 'vrw', fs)
            p.gyro = sensor_model(gyro_err, 'arw', fs)
            # vibration (Sim(env=...)): vib_def dicts as Sim.__parse_env makes them; the lane-per-run kernels of both precisions
            # and the time-parallel series kernels carry the term
            self._bufs = {}
            p.vib_accel = self._vibration(ctx, vib_accel, 0, float(fs), precision)
            p.vib_gyro = self._vibration(ctx, vib_gyro, 1, float(fs), precision)
        else:
            if vib_accel is not None or vib_gyro is not None:
                raise ValueError('given sensors: a vibration model cannot be added to sensor series that already exist')
            p.given_sensors = 1
            p.in_gyro = given['gyro'].ptr
            p.in_accel = given['accel'].ptr if 'accel' in given else None
            p.in_odo = given['odo'].ptr if 'odo' in given else None
            self._given = given         # keeps the buffers alive
        if given is None and 'odo' in self.algos and (odo_err is None or 'ref_odo' not in truth):
            raise ValueError('the odometer algorithm needs odo_err and truth["ref_odo"]')
        if self.want_odo:
            p.odo_scale, p.odo_stdv = float(odo_err['scale']), float(odo_err['stdv'])
        self._ref_nav = np.ascontiguousarray(np.concatenate([truth['ref_att'], truth['ref_pos'], truth['ref_vel']], axis=1))
        end = self._ref_nav[-1]
        p.ref_end[:] = [float(x) for x in end]
        # device-resident inputs
        # (one allocation and one copy for all of them: ini table, truth specific force / angular rate [/ forward speed])
        self._bufs = getattr(self, '_bufs', {})
        parts = [table.reshape(-1), np.asarray(truth['ref_accel'], dtype=np.float64).reshape(-1),
                 np.asarray(truth['ref_gyro'], dtype=np.float64).reshape(-1)]
        if self.want_odo:
            parts.append(np.asarray(truth['ref_odo'], dtype=np.float64).reshape(-1))
        offs = np.cumsum([0] + [q.size for q in parts]) * 8
        self._bufs['inputs'] = ctx.upload(np.concatenate(parts))
        p.ini, p.ref_accel, p.ref_gyro = (self._bufs['inputs'].at(offs[k]) for k in range(3))
        if self.want_odo:
            p.ref_odo = self._bufs['inputs'].at(offs[3])
        # per-run position origin for the fp32 displacement series (free_integration.py:96-98 / :127-128)
        self._ini_table, self._ini_first, self._ref_frame = table, int(ini_first), int(ref_frame)
        # outputs
        plane = self.n * self.runs * self._esize
        self.sensor_layout = 'runs'
        # the regions a launch streams at once (sensor series, every algorithm's trajectories): carved from the device's placed
        # arena when they are large -- ONE reservation for all of them, so that the arena grows (searches) once
        big = (6 * plane + (plane if self.want_odo else 0) if self.keep_sensors else 0) + (9 * plane * len(self.algos) if self.keep_traj else 0)
        use_placed = (big >= ctx.PLACED_MIN_JOB) if placed is None else bool(placed)
        use_placed = bool(use_placed and big > 0 and ctx.placed_reserve(big))
        if self.keep_sensors:
            if not self.algos and given is None and precision == 'f64' and _lib.VIB_PSD not in (p.vib_accel.type, p.vib_gyro.type):
                # few runs, long series: the time-parallel series kernels, series-major output (the library decides)
                p.sensor_layout = 1
                v = C.c_int32(0)
                check(lib.ginsim_mc_variant(C.byref(p), C.byref(v)))
                if v.value == 2:
                    self.sensor_layout = 'series'
                else:
                    p.sensor_layout = 0
            lay = self.sensor_layout
            # one allocation, accel then gyro: series-major (or one run) that IS the [sensor][run][axis][n] layout ginsim_allan reads
            self._bufs['imu'] = ctx.malloc(6 * plane, placed=use_placed)
            self._bufs['accel'] = DeviceView(self._bufs['imu'], 0, 3 * plane, lay)
            self._bufs['gyro'] = DeviceView(self._bufs['imu'], 3 * plane, 3 * plane, lay)
            p.out_accel, p.out_gyro = self._bufs['accel'].ptr, self._bufs['gyro'].ptr
            if self.want_odo:
                self._bufs['odo'] = ctx.malloc(plane, placed=use_placed)
                self._bufs['odo'].layout = lay
                p.out_odo = self._bufs['odo'].ptr
        self.proc_first, self.proc_ned, self.end_ned = proc_first, bool(proc_ned), bool(end_ned)
        if proc_first is not None:
            if len(self.algos) != 1 or given is not None or precision != 'f64':
                raise ValueError('online process statistics: one algorithm per job, generated sensors, fp64')
            if not 0 <= int(proc_first) < self.n:
                raise ValueError('proc_first must be a sample index of the run')
            if proc_ned and int(ref_frame) != 0:
                raise ValueError('NED position errors exist in ref_frame 0 only')
            self._bufs['ref_nav'] = ctx.upload(self._ref_nav)
            p.ref_nav, p.proc_first, p.proc_pos_ned = self._bufs['ref_nav'].ptr, int(proc_first), int(bool(proc_ned))
            p.proc_plain_sums = int(starts_on_truth(table, self._ref_nav[0], ref_frame))
        if end_ned and (int(ref_frame) != 0 or precision != 'f64' or given is not None):
            raise ValueError('end_ned: ref_frame 0, fp64, generated sensors')
        for a in self.algos:
            s = ALGO_SLOT[a]
            self._bufs['end_' + a] = ctx.malloc(9 * self.runs * 8)
            p.out_end[s] = self._bufs['end_' + a].ptr
            if end_ned:
                self._bufs['endned_' + a] = ctx.malloc(9 * self.runs * 8)
                p.out_end_ned[s] = self._bufs['endned_' + a].ptr
            if proc_first is not None:
                self._bufs['proc_' + a] = ctx.malloc(27 * self.runs * 8)
                p.out_proc[s] = self._bufs['proc_' + a].ptr
            if self.keep_traj:
                self._bufs['traj_' + a] = ctx.malloc(9 * plane, placed=use_placed)
                p.out_traj[s] = self._bufs['traj_' + a].ptr

    # bytes the launch writes to HBM (the algorithmic traffic of SURVEY 8(d))
    def allan(self, fs=None, names=('accel', 'gyro')):
        """Allan deviation of the kept sensor series on the device (allan_analysis.py:33-49 for every run at once):
        returns (tau (ntau,), {name: (runs, ntau, 3)}).  The series of all named sensors are laid out as
        [sensor][run][axis][n] (one run is already three contiguous series; more runs are re-laid out
        [3][n][runs] -> [runs][3][n] on the device) and go through ONE ginsim_allan call."""
        if not self.keep_sensors or self.precision != 'f64':
            raise ValueError('Allan analysis needs the fp64 sensor series (keep_sensors=True)')
        fs = float(self.params.fs if fs is None else fs)
        names = tuple(names)
        per = 3 * self.n * self.runs * 8
        contiguous = (self.runs == 1 or self.sensor_layout == 'series') and \
            all(self._bufs[names[i + 1]].ptr == self._bufs[names[i]].ptr + per for i in range(len(names) - 1))
        if contiguous:              # already [sensor][run][axis][n]: the Allan kernels read the job's own buffer
            ptr = self._bufs[names[0]].ptr
        else:
            # the re-laid-out copy lives with the job (released with it): a 2 GB hipMalloc + hipFree per call cost 0.4 ms,
            # as much as the Allan kernels themselves
            tmp = self._bufs.get('_allan_layout')
            if tmp is None or tmp.nbytes < per * len(names):
                if tmp is not None:
                    tmp.free()
                tmp = self._bufs['_allan_layout'] = self.ctx.malloc(per * len(names))
            for i, nm in enumerate(names):      # one run: C = 3, R = 1 makes the re-layout a plain copy
                if self.sensor_layout == 'series':
                    check(lib.ginsim_runs_to_series(self.ctx.handle, self._bufs[nm].ptr, 1, 3 * self.n * self.runs, 1, tmp.at(i * per)))
                else:
                    check(lib.ginsim_runs_to_series(self.ctx.handle, self._bufs[nm].ptr, 3, self.n, self.runs, tmp.at(i * per)))
            ptr = tmp.ptr
        avar, tau = allan_var(self.ctx, ptr, self.n, 3 * self.runs * len(names), self.n, fs)
        ad = np.sqrt(avar).reshape(len(names), self.runs, 3, -1)
        return tau, {nm: ad[i].transpose(0, 2, 1).copy() for i, nm in enumerate(names)}

    def buffer(self, name):
        """Device buffer of a materialised series ('accel', 'gyro', 'odo', 'traj_free', ...), e.g. to feed given=."""
        if name not in self._bufs:
            raise ValueError('%r was not kept by this job' % (name,))
        return self._bufs[name]

    def bytes_written(self):
        per_sample = 0
        if self.keep_sensors:
            per_sample += 6 * self._esize + (self._esize if self.want_odo else 0)
        if self.keep_traj:
            per_sample += 9 * self._esize * len(self.algos)
        return per_sample * self.n * self.runs + 72 * self.runs * len(self.algos)

    def launch(self):
        """Enqueue the fused kernel on the context's stream (asynchronous)."""
        check(self.ctx.retry_oom(lambda: lib.ginsim_mc_run(self.ctx.handle, C.byref(self.params))))

    def placement(self):
        """Where the materialised series of this job lie: {'placed': regions carved from the device's placed arena, 'bytes': their
        size, 'arena': Context.placed_info()} -- the launch streams all of them at once, and with its planes across the three
        classes of physical memory it writes 6.8-7.0 TB/s instead of the 5.7-5.9 TB/s of planes that hipMalloc put into one."""
        big = {k: b for k, b in self._bufs.items() if isinstance(b, DeviceBuffer) and (k == 'imu' or k == 'odo' or k.startswith('traj_'))}
        placed = sorted(k for k, b in big.items() if b.placed)
        return {'placed': placed, 'bytes': sum(big[k].nbytes for k in placed), 'unplaced': sorted(k for k in big if k not in placed),
                'arena': self.ctx.placed_info()}

    def kernel_name(self):
        """Name of the kernel launch() dispatches for these parameters (as rocprofv3 reports it, without arguments),
        reported by the library's own dispatch code (ginsim_mc_kernel_name)."""
        buf = C.create_string_buffer(256)
        check(lib.ginsim_mc_kernel_name(C.byref(self.params), buf, 256))
        return buf.value.decode()

    def run(self):
        self.launch()
        self.ctx.sync()
        return self

    def stats(self, algo, ned=False):
        """End-point statistics of the last launch; ned=True: position error in NED metres (needs end_ned=True)."""
        if ned and not self.end_ned:
            raise ValueError('the NED end-point record was not requested (end_ned=True)')
        s = _lib.Stats()
        check(lib.ginsim_end_stats(self.ctx.handle, self._bufs[('endned_' if ned else 'end_') + algo].ptr, self.runs, C.byref(s)))
        return StatsResult(s)

    def process_stats_online(self, algo):
        """(runs, 3, 9) = max|e|, mean, std of the error over samples >= proc_first, as the kernel accumulated them."""
        if self.proc_first is None:
            raise ValueError('online process statistics were not requested (proc_first=...)')
        # a (runs, 3, 9) VIEW of the [3][9][runs] device layout: at 262 144 runs the transposing copy took 50 ms of results()
        return self.ctx.download(self._bufs['proc_' + algo], (3, 9, self.runs)).transpose(2, 0, 1)

    def stats_begin(self, algo, slot=0):
        """Enqueue the end-point reduction of the last launch() into pinned slot 0..7 without waiting for it."""
        check(lib.ginsim_end_stats_begin(self.ctx.handle, self._bufs['end_' + algo].ptr, self.runs, int(slot)))

    def stats_finish(self, slot=0):
        """Wait for stats_begin(slot) only (later launches on the stream keep running) and return its record."""
        s = _lib.Stats()
        check(lib.ginsim_end_stats_finish(self.ctx.handle, int(slot), C.byref(s)))
        return StatsResult(s)

    def stats_all_begin(self, algo, slot=0):
        """stats_begin over ALL ranks (the context needs comm_init): reduction -> RCCL all-gather of the records -> pinned slot,
        enqueued on the stream without waiting."""
        check(lib.ginsim_end_stats_all_begin(self.ctx.handle, self._bufs['end_' + algo].ptr, self.runs, int(slot)))

    def stats_all_finish(self, slot=0):
        """Wait for stats_all_begin(slot) only; the merged record of every rank's runs (identical on every rank)."""
        s = _lib.Stats()
        check(lib.ginsim_end_stats_all_finish(self.ctx.handle, int(slot), C.byref(s)))
        return StatsResult(s)

    def process_stats(self, algo, first_sample=0, pos_ned=False):
        """Per-run statistics of the error over time (samples >= first_sample): (runs, 3, 9) = max|e|, mean, std.
        Needs the trajectories (keep_traj=True) and truth['ref_att'/'ref_pos'/'ref_vel']."""
        if not self.keep_traj:
            raise ValueError('process-error statistics need the trajectories (keep_traj=True)')
        if 'ref_nav' not in self._bufs:
            self._bufs['ref_nav'] = self.ctx.upload(self._ref_nav)
        out = np.empty((self.runs, 3, 9))
        if self.precision == 'f32':     # float series, positions as displacement from the run's initial position
            check(lib.ginsim_process_stats_f32(self.ctx.handle, self._bufs['traj_' + algo].ptr, self._bufs['ref_nav'].ptr, self.n,
                                               self.runs, int(first_sample), int(bool(pos_ned)), self._origin().ptr,
                                               self._ini_table.shape[0], self._ini_first, dptr(out)))
            return out
        check(lib.ginsim_process_stats(self.ctx.handle, self._bufs['traj_' + algo].ptr, self._bufs['ref_nav'].ptr,
                                       self.n, self.runs, int(first_sample), int(bool(pos_ned)), dptr(out)))
        return out

    def _origin(self):
        """Device table of the initial positions the fp32 displacement series are relative to ([n_ini][3]: ECEF for ref_frame 1,
        LLA for ref_frame 0; free_integration.py:96-98 / :127-128)."""
        if '_origin' not in self._bufs:
            from gnss_ins_sim.geoparams import geoparams
            lla = self._ini_table[:, 0:3]
            self._bufs['_origin'] = self.ctx.upload(np.ascontiguousarray(geoparams.lla2ecef(lla) if self._ref_frame == 1 else lla))
        return self._bufs['_origin']

    def stats_from_traj(self, algo, pos_ned=False):
        """End-point statistics recomputed on the device from the kept trajectories (used for extra_opt='ned')."""
        if not self.keep_traj:
            raise ValueError('needs the trajectories (keep_traj=True)')
        if 'ref_nav' not in self._bufs:
            self._bufs['ref_nav'] = self.ctx.upload(self._ref_nav)
        s = _lib.Stats()
        if self.precision == 'f32':
            check(lib.ginsim_end_stats_from_traj_f32(self.ctx.handle, self._bufs['traj_' + algo].ptr, self._bufs['ref_nav'].ptr, self.n,
                                                     self.runs, int(bool(pos_ned)), self._origin().ptr, self._ini_table.shape[0],
                                                     self._ini_first, C.byref(s)))
            return StatsResult(s)
        check(lib.ginsim_end_stats_from_traj(self.ctx.handle, self._bufs['traj_' + algo].ptr, self._bufs['ref_nav'].ptr,
                                             self.n, self.runs, int(bool(pos_ned)), C.byref(s)))
        return StatsResult(s)

    def end_errors(self, algo, ned=False):
        """(runs, 9) end-point errors [att3 wrapped, pos3, vel3]; ned=True: the NED record (end_ned=True)."""
        return self.ctx.download(self._bufs[('endned_' if ned else 'end_') + algo], (9, self.runs)).T.copy()

    def _gather(self, ptr, ncomp, run_ids, series_major=False):
        ids = np.ascontiguousarray(np.asarray(run_ids, dtype=np.int64).reshape(-1))
        out = np.empty((ids.size, self.n, ncomp))
        fn = lib.ginsim_gather_runs_f32 if self.precision == 'f32' else (lib.ginsim_gather_series if series_major else lib.ginsim_gather_runs)
        check(self.ctx.retry_oom(lambda: fn(self.ctx.handle, ptr, ncomp, self.n, self.runs, ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                            ids.size, dptr(out))))
        return out

    def sensors(self, name, run_ids):
        """Sensor series of selected runs: 'accel'/'gyro' -> (k,n,3); 'odo' -> (k,n)."""
        if not self.keep_sensors:
            raise ValueError('sensor series were not kept (keep_sensors=False)')
        sm = self.sensor_layout == 'series'
        if name == 'odo':
            return self._gather(self._bufs['odo'].ptr, 1, run_ids, sm)[:, :, 0]
        return self._gather(self._bufs[name].ptr, 3, run_ids, sm)

    def trajectories(self, algo, run_ids, displacement=False):
        """(att, pos, vel) of selected runs, each (k,n,3).  displacement=True (fp32 jobs): the position series as the kernel
        wrote it -- the displacement from the run's initial position -- instead of initial position + displacement."""
        if not self.keep_traj:
            raise ValueError('trajectories were not kept (keep_traj=False)')
        base = self._bufs['traj_' + algo].ptr
        plane = self.n * self.runs * self._esize
        att, pos, vel = (self._gather(base + 3 * k * plane, 3, run_ids) for k in range(3))
        if self.precision == 'f32' and not displacement:         # the device series is the displacement from the run's initial position
            from gnss_ins_sim.geoparams import geoparams
            ids = np.asarray(run_ids, dtype=np.int64).reshape(-1)
            for k, r in enumerate(ids):
                call = self._ini_first + int(r)
                ini = self._ini_table[call if call < self._ini_table.shape[0] else 0]
                pos[k] += geoparams.lla2ecef(ini[0:3]) if self._ref_frame == 1 else ini[0:3]
        return att, pos, vel

    def _vibration(self, ctx, vib_def, sensor, fs, precision):
        """ginsim_vibration of one sensor (0 accelerometer, 1 gyroscope).  A 'psd' definition becomes type 3: the series of all
        runs, [3][period][runs], are made on the device now (pathgen.py:479-484 / :541-546 make them per run)."""
        if vib_def is None or str(vib_def['type']).lower() != 'psd':
            return vibration(vib_def, fs, random_phase=bool(sensor))
        if precision != 'f64':
            raise NotImplementedError("the 'psd' vibration runs on the fp64 kernels only")
        v = _lib.Vibration()
        made = psd_amplitudes(vib_def, fs, self.n)
        if made is None:                # the reference's time_series_from_psd returns zeros (the PSD reaches beyond fs / 2)
            return v
        period, amp, on_grid = made
        buf = self._bufs['vib_psd_%d' % sensor] = ctx.malloc(3 * period * self.runs * 8)
        amp = np.ascontiguousarray(amp)
        p = self.params
        check(ctx.retry_oom(lambda: lib.ginsim_vib_psd_series(ctx.handle, amp.ctypes.data, period, self.runs, p.run_offset, p.seed,
                                                             sensor, int(on_grid), buf.ptr)))
        v.type, v.series, v.period = _lib.VIB_PSD, buf.ptr, period
        self.psd_given_on_grid = getattr(self, 'psd_given_on_grid', False) or on_grid
        return v

    def release(self):
        for b in self._bufs.values():
            b.free()
        self._bufs = {}


class AuxSensorJob(object):
    """GPS / magnetometer measurements of a Monte-Carlo batch (pathgen.gps_gen, mag_gen), kept in HBM as
    gps[6][m][runs] and mag[3][n][runs].  Same Philox streams as the reference-injection shim (oracle/ref_shim.py)."""

    def __init__(self, ctx, runs, seed=0, run_offset=0, ref_gps=None, gps_err=None, ref_frame=0, ref_mag=None,
                 mag_err=None):
        from gnss_ins_sim.geoparams import geoparams
        self.ctx, self.runs = ctx, int(runs)
        p = self.params = _lib.AuxParams()
        p.runs, p.run_offset, p.seed = self.runs, int(run_offset), int(seed) & (2 ** 64 - 1)
        self._bufs = {}
        self.m = self.n = 0
        if ref_gps is not None:
            ref_gps = np.ascontiguousarray(ref_gps, dtype=np.float64)
            self.m = p.m = ref_gps.shape[0]
            sig = np.concatenate([np.asarray(gps_err['stdp'], dtype=np.float64) * np.ones(3),
                                  np.asarray(gps_err['stdv'], dtype=np.float64) * np.ones(3)])
            if ref_frame == 0:                       # metres -> rad at the FIRST GPS point, pathgen.py:616-619
                rm, rn, _, _, cl, _ = geoparams.geo_param(ref_gps[0, 0:3])
                sig[0] = sig[0] / rm
                sig[1] = sig[1] / rn / cl
            p.gps_sigma[:] = [float(x) for x in sig]
            self._bufs['ref_gps'] = ctx.upload(ref_gps)
            self._bufs['gps'] = ctx.malloc(6 * self.m * self.runs * 8)
            p.ref_gps, p.out_gps = self._bufs['ref_gps'].ptr, self._bufs['gps'].ptr
        if ref_mag is not None:
            ref_mag = np.ascontiguousarray(ref_mag, dtype=np.float64)
            self.n = p.n = ref_mag.shape[0]
            p.mag_si[:] = [float(x) for x in np.asarray(mag_err['si'], dtype=np.float64).reshape(9)]
            p.mag_hi[:] = [float(x) for x in np.asarray(mag_err['hi'], dtype=np.float64) * np.ones(3)]
            p.mag_std[:] = [float(x) for x in np.asarray(mag_err['std'], dtype=np.float64) * np.ones(3)]
            self._bufs['ref_mag'] = ctx.upload(ref_mag)
            self._bufs['mag'] = ctx.malloc(3 * self.n * self.runs * 8)
            p.ref_mag, p.out_mag = self._bufs['ref_mag'].ptr, self._bufs['mag'].ptr

    def run(self):
        check(lib.ginsim_aux_sensors(self.ctx.handle, C.byref(self.params)))
        self.ctx.sync()
        return self

    def series(self, name, run_ids):
        """'gps' -> (k, m, 6); 'mag' -> (k, n, 3)."""
        ids = np.ascontiguousarray(np.asarray(run_ids, dtype=np.int64).reshape(-1))
        ncomp, length = (6, self.m) if name == 'gps' else (3, self.n)
        out = np.empty((ids.size, length, ncomp))
        check(self.ctx.retry_oom(lambda: lib.ginsim_gather_runs(self.ctx.handle, self._bufs[name].ptr, ncomp, length, self.runs,
                                                                ids.ctypes.data_as(C.POINTER(C.c_int64)), ids.size, dptr(out))))
        return out

    def release(self):
        for b in self._bufs.values():
            b.free()
        self._bufs = {}


def pinned_empty(ctx, shape, dtype=np.float64):
    """An uninitialised NumPy array in page-locked host memory (ginsim_host_alloc): host-buffer calls on such arrays copy
    at the link rate.  The memory is released when the array (and every view of it) is gone."""
    import weakref
    shape = tuple(int(v) for v in np.atleast_1d(shape))
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    check(lib.ginsim_host_alloc(ctx.handle, max(nbytes, 8), C.byref(p)))
    raw = (C.c_char * max(nbytes, 8)).from_address(p.value)
    arr = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    addr = p.value

    def release(owner=ctx):         # the closure keeps the Context object (not its raw handle) alive as long as the pages
        lib.ginsim_host_free(owner.handle, C.c_void_p(addr))     # handle is None once the context was closed: plain hipHostFree
    weakref.finalize(raw, release)
    return arr


def free_integration_host(ctx, algo, ref_frame, fs, gyro, accel=None, odo=None, ini=None, earth_rot=True,
                          ini_first=0, pinned_out=False, out=None):
    """Given-data mechanisation through ginsim_free_integration (host buffers in, host buffers out).
    gyro/accel (R,n,3) or (n,3); odo (R,n) or (n,).  Returns att, pos, vel with gyro's leading shape.
    pinned_out: the three results in page-locked memory (pinned_empty); inputs made with pinned_empty are used as they are.
    out: (att, pos, vel) arrays of shape (R, n, 3) to write into (e.g. page-locked ones kept across calls: locking pages costs
    about what it saves on ONE copy)."""
    g = np.asarray(gyro, dtype=np.float64)
    single = g.ndim == 2
    g = np.ascontiguousarray(g.reshape((-1,) + g.shape[-2:]))
    R, n, _ = g.shape
    a = None if accel is None else np.ascontiguousarray(np.asarray(accel, dtype=np.float64).reshape(R, n, 3))
    o = None if odo is None else np.ascontiguousarray(np.asarray(odo, dtype=np.float64).reshape(R, n))
    table, has_g = ini_table(ini)
    if out is not None:
        att, pos, vel = out
        for v in out:
            if v.shape != (R, n, 3) or v.dtype != np.float64 or not v.flags['C_CONTIGUOUS']:
                raise ValueError('out: three C-contiguous float64 arrays of shape (%d, %d, 3)' % (R, n))
    else:
        new = (lambda: pinned_empty(ctx, (R, n, 3))) if pinned_out else (lambda: np.empty((R, n, 3)))
        att, pos, vel = new(), new(), new()
    check(ctx.retry_oom(lambda: lib.ginsim_free_integration(ctx.handle, ALGO_BITS[algo], int(ref_frame), float(fs), int(bool(earth_rot)),
                                                            dptr(g), dptr(a), dptr(o), R, n, dptr(table), table.shape[0], int(has_g),
                                                            int(ini_first), dptr(att), dptr(pos), dptr(vel))))
    if single:
        return att[0], pos[0], vel[0]
    return att, pos, vel


def rng_normals(ctx, seed, run, stream, count, words=False):
    z0, z1 = np.empty(count), np.empty(count)
    w = np.empty((count, 4), dtype=np.uint32) if words else None
    check(lib.ginsim_rng_normals(ctx.handle, int(seed), int(run), int(stream), int(count), dptr(z0), dptr(z1),
                                 None if w is None else w.ctypes.data_as(C.POINTER(C.c_uint32))))
    return (z0, z1, w) if words else (z0, z1)


def normal_transform(ctx, words):
    """The device's normal transform on given Philox words (count, 4) uint32 -> (z0 from word 0, z1 from word 1).  Test hook
    for corner cases."""
    w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32).reshape(-1, 4))
    z0, z1 = np.empty(w.shape[0]), np.empty(w.shape[0])
    check(lib.ginsim_normal_transform(ctx.handle, w.ctypes.data_as(C.POINTER(C.c_uint32)), w.shape[0], dptr(z0), dptr(z1)))
    return z0, z1


def allan_var(ctx, x, n, nseries, series_stride, fs, cap=128):
    """Allan variance of `nseries` device-resident series (DeviceBuffer, DeviceView or raw pointer), allan.py:18-59.
    Returns (avar (nseries, ntau), tau (ntau,))."""
    ptr = getattr(x, 'ptr', x)
    tau = np.empty(cap)                 # the library writes every entry it reports (ntau of them per series)
    avar = np.empty((nseries, cap))
    nt = C.c_int32(0)
    check(ctx.retry_oom(lambda: lib.ginsim_allan(ctx.handle, ptr, int(n), int(nseries), int(series_stride), float(fs), dptr(tau),
                                                 dptr(avar), C.byref(nt), cap)))
    return np.ascontiguousarray(avar[:, :nt.value]), tau[:nt.value].copy()


def allan_var_host(ctx, series, fs):
    """Host arrays in: series (n,) or (S, n).  Uploads, runs the device kernels, returns (avar, tau)."""
    a = np.ascontiguousarray(np.atleast_2d(np.asarray(series, dtype=np.float64)))
    buf = ctx.upload(a)
    try:
        avar, tau = allan_var(ctx, buf, a.shape[1], a.shape[0], a.shape[1], fs)
    finally:
        buf.free()
    return (avar[0], tau) if np.ndim(series) == 1 else (avar, tau)
