"""Data registry + error statistics, interface-compatible with the reference's ``InsDataMgr``
(gnss_ins_sim/sim/ins_data_manager.py): the same ~35 named series with the same units/legends
(:44-273), ``add_data / get_data / get_data_all / get_data_properties / set_algo_output / is_supported /
is_available / available``, and ``get_error_stats`` returning {'max','avg','std','units'}.

Difference in kind, not in interface: Monte-Carlo series are ``McSeries`` views over GPU buffers and the
end-point statistics (:717-759, 797-808) are reduced ON THE DEVICE; nothing here loops over runs.
"""
import numpy as np

from . import sim_data
from .sim_data import Sim_data

_XYZ = ['m', 'm', 'm']
_LLA = ['rad', 'rad', 'm']
_LLA_OUT = ['deg', 'deg', 'm']
_RATE = ['rad/s'] * 3
_RATE_OUT = ['deg/s'] * 3
_ACC = ['m/s^2'] * 3
_VEL = ['m/s'] * 3
_ANG = ['rad'] * 3
_ANG_OUT = ['deg'] * 3
_UT = ['uT'] * 3

# name, description, units, output_units, legend, extra kwargs   (ins_data_manager.py:44-231)
_REGISTRY = [
    ('fs', 'Sample frequency of IMU', ['Hz'], None, None, {'plottable': False}),
    ('fs_gps', 'Sample frequency of GPS', ['Hz'], None, None, {'plottable': False}),
    ('fs_mag', 'Sample frequency of Magnetometer', ['Hz'], None, None, {'plottable': False}),
    ('ref_frame', 'Reference frame', None, None, None, {'plottable': False}),
    ('time', 'sample time', ['sec'], None, ['time'], {}),
    ('gps_time', 'GPS sample time', ['sec'], None, ['gps_time'], {}),
    ('gps_visibility', 'GPS visibility', None, None, ['gps_visibility'], {}),
    ('ref_pos', 'true LLA pos in the navigation frame', _LLA, _LLA_OUT, ['ref_pos_lat', 'ref_pos_lon', 'ref_pos_alt'], {}),
    ('ref_vel', 'true vel in the NED frame', _VEL, None, ['ref_vel_x', 'ref_vel_y', 'ref_vel_z'], {}),
    ('ref_att_euler', 'true attitude (Euler angles, ZYX)', _ANG, _ANG_OUT, ['ref_Yaw', 'ref_Pitch', 'ref_Roll'], {}),
    ('ref_att_quat', 'true attitude (quaternion)', None, None, ['q0', 'q1', 'q2', 'q3'], {}),
    ('ref_gyro', 'true angular velocity in the body frame', _RATE, _RATE_OUT, ['ref_gyro_x', 'ref_gyro_y', 'ref_gyro_z'], {}),
    ('ref_accel', 'true accel in the body frame', _ACC, None, ['ref_accel_x', 'ref_accel_y', 'ref_accel_z'], {}),
    ('ref_gps', 'true GPS LLA position and NED velocity', _LLA + _VEL, _LLA_OUT + _VEL,
     ['ref_gps_lat', 'ref_gps_lon', 'ref_gps_alt', 'ref_gps_vN', 'ref_gps_vE', 'ref_gps_vD'], {}),
    ('ref_odo', 'true odometer velocity', ['m/s'], None, ['ref_odo'], {}),
    ('ref_mag', 'true magnetic field in the body frame', _UT, None, ['ref_mag_x', 'ref_mag_y', 'ref_mag_z'], {}),
    ('gyro', 'gyro measurements', _RATE, _RATE_OUT, ['gyro_x', 'gyro_y', 'gyro_z'], {}),
    ('accel', 'accel measurements', _ACC, None, ['accel_x', 'accel_y', 'accel_z'], {}),
    ('gps', 'GPS LLA position and NED velocity measurements', _LLA + _VEL, _LLA_OUT + _VEL,
     ['gps_lat', 'gps_lon', 'gps_alt', 'gps_vN', 'gps_vE', 'gps_vD'], {}),
    ('odo', 'odometer velocity measurement', ['m/s'], None, ['odo'], {}),
    ('mag', 'magnetometer measurements', _UT, None, ['mag_x', 'mag_y', 'mag_z'], {}),
    ('gyro_cal', 'gyro measurements after factory calibration', _RATE, _RATE_OUT, ['gyro_x', 'gyro_y', 'gyro_z'], {}),
    ('accel_cal', 'accel measurements after factory calibration', _ACC, None, ['accel_x', 'accel_y', 'accel_z'], {}),
    ('mag_cal', 'magnetometer measurements after SI&HI calibration', _UT, None, ['mag_x', 'mag_y', 'mag_z'], {}),
    ('soft_iron', 'soft iron calibration matrix', None, None, None, {'plottable': False}),
    ('hard_iron', 'hard iron', ['uT'] * 4, None, ['offset_x', 'offset_y', 'offset_z', 'radius'], {'plottable': False}),
    ('algo_time', 'sample time from algo', ['sec'], None, None, {}),
    ('pos', 'simulation position from algo', _LLA, _LLA_OUT, ['pos_lat', 'pos_lon', 'pos_alt'], {}),
    ('vel', 'simulation velocity from algo', _VEL, None, ['vel_x', 'vel_y', 'vel_z'], {}),
    ('att_quat', 'simulation attitude (quaternion) from algo', None, None, ['q0', 'q1', 'q2', 'q3'], {}),
    ('att_euler', 'simulation attitude (Euler, ZYX) from algo', _ANG, _ANG_OUT, ['Yaw', 'Pitch', 'Roll'], {}),
    ('wb', 'gyro bias estimation', _RATE, _RATE_OUT, ['gyro_bias_x', 'gyro_bias_y', 'gyro_bias_z'], {}),
    ('ab', 'accel bias estimation', _ACC, None, ['accel_bias_x', 'accel_bias_y', 'accel_bias_z'], {}),
    ('ad_gyro', 'Allan deviation of gyro', _RATE, _RATE_OUT, ['AD_wx', 'AD_wy', 'AD_wz'], {'logx': True, 'logy': True}),
    ('ad_accel', 'Allan deviation of accel', _ACC, None, ['AD_ax', 'AD_ay', 'AD_az'], {'logx': True, 'logy': True}),
]

# which slice of the 9-component end-point record each error quantity occupies
_END_SLICE = {'att_euler': slice(0, 3), 'pos': slice(3, 6), 'vel': slice(6, 9)}


class InsDataMgr(object):
    def __init__(self, fs, ref_frame=0):
        self._all = {}
        for name, desc, units, out_units, legend, extra in _REGISTRY:
            sd = Sim_data(name=name, description=desc, units=units, output_units=out_units, legend=legend, **extra)
            self._all[name] = sd
            setattr(self, name, sd)
        self.ref_frame.data = ref_frame if ref_frame in (0, 1) else 0
        if self.ref_frame.data == 1:                       # ins_data_manager.py:235-256
            for s, tag in ((self.ref_pos, 'ref_pos'), (self.pos, 'pos')):
                s.units, s.output_units = list(_XYZ), list(_XYZ)
                s.legend = [tag + '_x', tag + '_y', tag + '_z']
            self.ref_pos.description = 'true position in the local NED frame'
            for s, tag in ((self.ref_gps, 'ref_gps'), (self.gps, 'gps')):
                s.units, s.output_units = _XYZ + _VEL, _XYZ + _VEL
                s.legend = [tag + '_' + c for c in ('x', 'y', 'z', 'vx', 'vy', 'vz')]
            self.ref_gps.description = 'true GPS position and velocity in the local NED frame'
            self.gps.description = 'GPS position and velocity measurements in the local NED frame'
        self.available = [self.ref_frame.name]
        if fs[0] is None:
            raise ValueError('IMU sampling frequency cannot be None.')
        for sd, v in ((self.fs, fs[0]), (self.fs_gps, fs[1]), (self.fs_mag, fs[2])):
            if v is not None:
                sd.data = v
                self.available.append(sd.name)
        self._do_not_save = ['fs', 'fs_gps', 'fs_mag', 'ref_frame']
        self._algo_output = []
        self._mc = None         # device results of the last Sim.run: see set_mc_results

    # ------------------------------------------------------------------ registry
    def add_data(self, data_name, data, key=None, units=None):
        if data_name not in self._all:
            raise ValueError("Unsupported data: %s." % data_name)
        self._all[data_name].add_data(data, key, units)
        if data_name not in self.available:
            self.available.append(data_name)

    def set_algo_output(self, algo_output):
        for i in algo_output:
            if not self.is_supported(i):
                raise ValueError("Unsupported algorithm output: %s." % i)
            self._algo_output.append(i)

    def get_data(self, data_names):
        data = []
        for i in data_names:
            if i not in self.available:
                print('%s is not available.' % i)
                return None
            data.append(self._all[i].data)
        return data

    def get_data_all(self, data_name):
        return self._all.get(data_name)

    def get_data_properties(self, data_name):
        d = self._all[data_name]
        return [d.description, d.units, d.plottable, d.logx, d.logy, d.legend]

    def is_supported(self, data_name):
        return data_name in self._all

    def is_available(self, data_name, key=None):
        ok = data_name in self.available
        if ok and key is not None:
            d = self._all[data_name].data
            ok = hasattr(d, 'keys') and key in d
        return ok

    # ------------------------------------------------------------------ Monte-Carlo results on the device
    def set_mc_results(self, mc):
        """mc: object with .algo_names (list), .end_stats(algo_name) -> ginsim.StatsResult (already merged
        across ranks), .process_stats(algo_name, data_name, start_index) (optional)."""
        self._mc = mc

    def get_error_stats(self, data_name, err_stats_start=0, angle=False, use_output_units=False, extra_opt=''):
        """InsDataMgr.get_error_stats (ins_data_manager.py:385-452).

        Outputs of plugins inside the fused kernel are reduced on the device (their series may not even be kept);
        outputs of hosted plugins (user code, host arrays) and logged data get the reference's host computation
        (array_error :519-553, __end_point_error_stats :717-759, __process_error_stats :761-795).  When both kinds
        produced `data_name`, the groups are reported side by side as the reference does (:810-832)."""
        if data_name not in self.available:
            print('error stats: %s is not available.' % data_name)
            return None
        if 'ref_' + data_name not in self.available:
            print('%s has no reference.' % data_name)
            return None
        src = self._all[data_name]
        units, out_units = list(src.units), list(src.output_units)
        ned = data_name == 'pos' and self.ref_frame.data == 0 and extra_opt == 'ned'
        if ned:
            units, out_units = list(_XYZ), list(_XYZ)
        on_device = self._mc is not None and data_name in _END_SLICE
        names = list(self._mc.algo_names) if on_device else []
        host = self._host_part(src.data, names)                 # {key: array} (or {None: array}) computed on the host
        end_point = err_stats_start == -1
        stat = None
        if names:
            if end_point:                                        # end-point statistics, :717-759
                per_algo = {a: self._mc.end_stats(a, ned=ned) for a in names}
                sl = _END_SLICE[data_name]
                pick = lambda st: {'max': st.maxabs[sl].copy(), 'avg': st.mean[sl].copy(), 'std': st.std[sl].copy()}
                stat = {s: {a: pick(per_algo[a])[s] for a in names} for s in ('max', 'avg', 'std')}
            else:                                                # process error of every run, :761-795
                t = np.asarray(self.time.data)
                hit = np.where(t >= max(err_stats_start, 0))[0]
                if hit.shape[0] == 0:
                    print('err_stats_start exceeds max data points.')
                stat = self._mc.process_stats(data_name, int(hit[0]) if hit.shape[0] else 0, ned=ned)
        if host:
            hstat = self._host_error_stats(data_name, host, err_stats_start, angle, ned)
            if hstat is None:
                return None
            if stat is None:
                stat = hstat
            else:
                for s in ('max', 'avg', 'std'):
                    stat[s].update(hstat[s] if isinstance(hstat[s], dict) else {'data': hstat[s]})
        if stat is None:
            return None
        if end_point and isinstance(stat['max'], dict) and len(stat['max']) == 1:     # one group: plain arrays, :727-735
            stat = {s: next(iter(stat[s].values())) for s in ('max', 'avg', 'std')}
        if use_output_units:
            for s in stat:
                if isinstance(stat[s], sim_data.RunStats):
                    stat[s] = stat[s].scaled(sim_data.unit_conversion_scale(units, out_units))
                elif isinstance(stat[s], dict):
                    stat[s] = {k: sim_data.convert_unit(v, units, out_units) for k, v in stat[s].items()}
                else:
                    stat[s] = sim_data.convert_unit(stat[s], units, out_units)
        stat['units'] = str(out_units)
        return stat

    @staticmethod
    def _host_part(data, device_names):
        """The host-resident part of a series: everything that is not a device view of a fused plugin."""
        if isinstance(data, sim_data.McSeries):
            return {}
        if isinstance(data, sim_data.ChainSeries):
            return dict(data.extra)
        if isinstance(data, dict):                              # a plain dict only ever holds host arrays
            return dict(data)
        if isinstance(data, np.ndarray):
            return {None: data}
        return {}

    def array_error(self, x, r, angle=False, lla=0):
        """InsDataMgr.array_error (ins_data_manager.py:519-553) for host arrays."""
        from ..attitude import attitude
        from ..geoparams import geoparams
        x, r = np.asarray(x, dtype=np.float64), np.asarray(r, dtype=np.float64)
        if lla == 0:
            err = x - r
            return attitude.angle_range_pi(err) if angle else err
        err = geoparams.lla2ecef_batch(x) - geoparams.lla2ecef_batch(r)
        if lla == 1:                                            # attitude.ecef_to_ned, attitude.py:596-603
            sl, cl, so, co = np.sin(r[:, 0]), np.cos(r[:, 0]), np.sin(r[:, 1]), np.cos(r[:, 1])
            err = np.stack([-sl * co * err[:, 0] - sl * so * err[:, 1] + cl * err[:, 2],
                            -so * err[:, 0] + co * err[:, 1],
                            -cl * co * err[:, 0] - cl * so * err[:, 1] - sl * err[:, 2]], axis=1)
        return err

    def calc_data_err(self, data_name, ref_data_name, angle=False, err_opt=''):
        """InsDataMgr.calc_data_err (ins_data_manager.py:454-517): the error SERIES of `data_name` against `ref_data_name` as a
        Sim_data 'err_<name>' -- array_error per key (angles wrapped to [-pi, pi]; err_opt 'ned' / 'ecef' for LLA positions in
        ref_frame 0), the reference interpolated onto algo_time where the lengths differ.  For Monte-Carlo series that live on
        the GPU the result is a lazy mapping: the error of a run is formed when that run is read (the statistics do not come
        this way: get_error_stats reduces on the device)."""
        if data_name not in self.available or ref_data_name not in self.available:
            print('%s or %s is not available.' % (data_name, ref_data_name))
            return None
        src = self._all[data_name]
        err = sim_data.Sim_data(name='err_' + data_name, description='ERROR of ' + src.description, units=src.units,
                                output_units=src.output_units, plottable=src.plottable, logx=src.logx, logy=src.logy,
                                grid=src.grid, legend=src.legend)
        lla = 0
        if data_name == self.pos.name and self.ref_frame.data == 0 and err_opt in ('ned', 'ecef'):
            lla = 1 if err_opt == 'ned' else 2
            err.description = 'ERROR of NED position' if lla == 1 else 'ERROR of ECEF position'
            err.units, err.output_units = list(_XYZ), list(_XYZ)
            err.legend = ['pos_N', 'pos_E', 'pos_D'] if lla == 1 else ['pos_x', 'pos_y', 'pos_z']
        ref_all = np.asarray(self._all[ref_data_name].data)
        t = np.asarray(self.time.data) if 'time' in self.available else None

        def one(key, x):
            x = np.asarray(x, dtype=np.float64)
            ref = ref_all
            if ref.shape[0] != x.shape[0]:                      # :489-497, 835-849
                at = self.algo_time.data if 'algo_time' in self.available else None
                at = at.get(key) if hasattr(at, 'get') else at
                if at is None or t is None:
                    raise ValueError('%s or %s is not available.' % (self.algo_time.name, self.time.name))
                tk = np.asarray(at, dtype=np.float64)
                ref = np.interp(tk, t, ref) if ref.ndim == 1 else np.stack([np.interp(tk, t, ref[:, c]) for c in range(ref.shape[1])], 1)
            return self.array_error(x, ref, angle, lla)

        data = src.data
        try:
            if isinstance(data, (sim_data.McSeries, sim_data.ChainSeries)):
                err.data = sim_data.DerivedSeries(data, one)
            elif isinstance(data, dict):
                err.data = {k: one(k, v) for k, v in data.items()}
            elif isinstance(data, np.ndarray):
                err.data = one(None, data)
        except ValueError as e:
            print(e)
            return None
        return err

    def _host_error_stats(self, data_name, data, err_stats_start, angle, ned):
        """Reference statistics of host arrays: data = {key: (n,k) array} or {None: array}."""
        ref_all = np.asarray(self._all['ref_' + data_name].data)
        t = np.asarray(self.time.data) if 'time' in self.available else None
        errs, starts = {}, {}
        for k, x in data.items():
            x = np.asarray(x, dtype=np.float64)
            ref, tk = ref_all, t
            if ref.shape[0] != x.shape[0]:                      # interpolate the reference, :489-497, 835-849
                at = self.algo_time.data if 'algo_time' in self.available else None
                at = at.get(k) if isinstance(at, dict) else at
                if at is None or t is None:
                    print('%s or %s is not available.' % (self.algo_time.name, self.time.name))
                    return None
                tk = np.asarray(at, dtype=np.float64)
                ref = np.interp(tk, t, ref) if ref.ndim == 1 else np.stack([np.interp(tk, t, ref[:, c]) for c in range(ref.shape[1])], 1)
            errs[k] = self.array_error(x, ref, angle, 1 if ned else 0)
            idx = 0
            if err_stats_start != -1 and tk is not None:
                hit = np.where(tk >= err_stats_start)[0]
                if hit.shape[0] == 0:
                    print('err_stats_start exceeds max data points.')
                idx = int(hit[0]) if hit.shape[0] else 0
            starts[k] = idx
        arr_stats = lambda e: {'max': np.max(np.abs(e), 0), 'avg': np.average(e, 0), 'std': np.std(e, 0)}
        if None in errs:                                        # a plain array: statistics over time, :752-754 / :790-793
            e = errs[None]
            return arr_stats(e[-1, :] if err_stats_start == -1 else e)
        if err_stats_start == -1:                               # end points of every key, grouped by '<group>_<idx>'
            groups = {}
            for k, e in errs.items():
                g = k.rpartition('_')[0] if isinstance(k, str) and '_' in k else 'data'
                groups.setdefault(g, []).append(e[-1, :] if e.ndim > 1 else e[-1:])
            per = {g: arr_stats(np.array(v)) for g, v in groups.items()}
            return {s: {g: per[g][s] for g in per} for s in ('max', 'avg', 'std')}
        stat = {'max': {}, 'avg': {}, 'std': {}}
        for k, e in errs.items():
            tmp = arr_stats(e[starts[k]:])
            for s in stat:
                stat[s][k] = tmp[s]
        return stat

    # ------------------------------------------------------------------ files
    def save_data(self, data_dir, max_runs=None):
        """InsDataMgr.save_data: every available series to '<name>[-<key>].csv'.  For Monte-Carlo series only
        runs below ``max_runs`` are written (None = all) -- a million CSV files help nobody.  The run id is read off
        the key itself (an int, or the suffix after the last '_' of '<algo>_<run>')."""
        def run_of(key):
            if isinstance(key, (int, np.integer)):
                return int(key)
            tail = str(key).rpartition('_')[2]
            return int(tail) if tail.isdigit() else 0

        saved = []
        for name in self.available:
            if name in self._do_not_save:
                continue
            sd = self._all[name]
            keys = None
            if isinstance(sd.data, (sim_data.McSeries, sim_data.ChainSeries)) and max_runs is not None:
                keys = [k for k in sd.data if run_of(k) < max_runs]
            sd.save_to_file(data_dir, keys)
            saved.append(name)
        return saved

    def save_kml_files(self, data_dir):
        print('KML export is outside the accelerated hot path (SURVEY.md section 2, #16); skipped.')

    def plot(self, *args, **kwargs):
        raise NotImplementedError('plotting is outside the accelerated hot path (SURVEY.md section 2, #17)')

    def show_plot(self):
        pass
