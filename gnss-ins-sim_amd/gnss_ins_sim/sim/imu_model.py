"""IMU error-model container, interface-compatible with the reference's ``imu_model.IMU``
(gnss_ins_sim/sim/imu_model.py:62-352): same constructor arguments, same attribute names
(``gyro_err``, ``accel_err``, ``mag_err``, ``gps_err``, ``odo_err``, ``gps``, ``odo``, ``magnetometer``),
same unit conversions (:138-143) and the same exceptions for bad input.

Deliberate difference: every IMU owns copies of its error dicts.  The reference aliases the module-level
profile dicts (imu_model.py:110-111), so a custom ``accuracy`` dict silently rewrites 'low-accuracy' for
every later IMU in the process.
"""
import copy
import math

import numpy as np

D2R = math.pi / 180


def _grade(gyro_drift_dph, arw_dprh, acc_drift, vrw_mpsrh, mag_std):
    return ({'b': np.zeros(3), 'b_drift': np.full(3, gyro_drift_dph) * D2R / 3600.0,
             'b_corr': np.full(3, 100.0), 'arw': np.full(3, arw_dprh) * D2R / 60.0},
            {'b': np.zeros(3), 'b_drift': np.full(3, acc_drift), 'b_corr': np.full(3, 100.0),
             'vrw': np.full(3, vrw_mpsrh) / 60.0},
            {'si': np.eye(3), 'hi': np.zeros(3), 'std': np.full(3, mag_std)})


# built-in grades, values of imu_model.py:18-52
gyro_low_accuracy, accel_low_accuracy, mag_low_accuracy = _grade(10.0, 0.75, 2.0e-4, 0.05, 0.1)
gyro_mid_accuracy, accel_mid_accuracy, mag_mid_accuracy = _grade(3.5, 0.25, 5.0e-5, 0.03, 0.01)
gyro_high_accuracy, accel_high_accuracy, mag_high_accuracy = _grade(0.1, 2.0e-3, 3.6e-6, 2.5e-5, 0.001)
gps_low_accuracy = {'stdp': np.array([5.0, 5.0, 7.0]), 'stdv': np.array([0.05, 0.05, 0.05])}    # :56-57
odo_low_accuracy = {'scale': 0.99, 'stdv': 0.1}                                                   # :60-61

_GRADES = {'low-accuracy': (gyro_low_accuracy, accel_low_accuracy, mag_low_accuracy),
           'mid-accuracy': (gyro_mid_accuracy, accel_mid_accuracy, mag_mid_accuracy),
           'high-accuracy': (gyro_high_accuracy, accel_high_accuracy, mag_high_accuracy)}
_REQUIRED = ('gyro_b', 'gyro_b_stability', 'gyro_arw', 'accel_b', 'accel_b_stability', 'accel_vrw')


class IMU(object):
    def __init__(self, accuracy='low-accuracy', axis=6, gps=True, gps_opt=None, odo=False, odo_opt=None):
        if axis == 9:
            self.magnetometer = True
        elif axis == 6:
            self.magnetometer = False
        else:
            raise ValueError('axis should be either 6 or 9.')
        self.gyro_err, self.accel_err, self.mag_err = copy.deepcopy(_GRADES['low-accuracy'])
        if isinstance(accuracy, str):
            if accuracy not in _GRADES:
                raise ValueError('accuracy is not a valid string.')
            self.gyro_err, self.accel_err, self.mag_err = copy.deepcopy(_GRADES[accuracy])
        elif isinstance(accuracy, dict):
            if not all(k in accuracy for k in _REQUIRED):
                raise ValueError('accuracy should at least have keys: \n' +
                                 'gyro_b, gyro_b_stability, gyro_arw, ' +
                                 'accel_b, accel_b_stability and accel_vrw')
            inf3 = np.array([float('inf')] * 3)
            self.gyro_err['b'] = accuracy['gyro_b'] * D2R / 3600.0               # deg/hr -> rad/s
            self.gyro_err['b_drift'] = accuracy['gyro_b_stability'] * D2R / 3600.0
            self.gyro_err['arw'] = accuracy['gyro_arw'] * D2R / 60.0              # deg/rt-hr -> rad/s/rt-Hz
            self.accel_err['b'] = accuracy['accel_b']
            self.accel_err['b_drift'] = accuracy['accel_b_stability']
            self.accel_err['vrw'] = accuracy['accel_vrw'] / 60.0                 # m/s/rt-hr -> m/s2/rt-Hz
            if self.magnetometer:
                if 'mag_std' not in accuracy:
                    raise ValueError('Magnetometer is enabled, but its noise std is not specified.')
                self.mag_err['std'] = accuracy['mag_std']
            self.gyro_err['b_corr'] = accuracy.get('gyro_b_corr', inf3)
            self.accel_err['b_corr'] = accuracy.get('accel_b_corr', inf3.copy())
            self.mag_err['si'] = accuracy.get('mag_si', np.eye(3))
            self.mag_err['hi'] = accuracy.get('mag_hi', np.zeros(3))
        else:
            raise TypeError('accuracy is not valid.')
        self.gps, self.gps_err = bool(gps), None
        if gps:
            self.gps_err = self._opt(gps_opt, gps_low_accuracy, ('stdp', 'stdv'), 'gps_opt')
        self.odo, self.odo_err = bool(odo), None
        if odo:
            self.odo_err = self._opt(odo_opt, odo_low_accuracy, ('scale', 'stdv'), 'odo_opt')

    @staticmethod
    def _opt(opt, default, keys, what):
        if opt is None:
            return copy.deepcopy(default)
        if not isinstance(opt, dict):
            raise TypeError('%s should be None or a dict' % what)
        if not all(k in opt for k in keys):
            raise ValueError('%s should have key: %s and %s' % (what, keys[0], keys[1]))
        return opt

    @staticmethod
    def _set(target, error, grades_index, what):
        if isinstance(error, str):
            if error not in _GRADES:
                raise ValueError('%s is not a valid string.' % what)
            return copy.deepcopy(_GRADES[error][grades_index])
        if isinstance(error, dict):
            for k in error:
                if k not in target:
                    raise ValueError('unsupported key: %s in %s' % (k, what))
                target[k] = error[k]
            return target
        raise TypeError('%s is not valid.' % what)

    def set_gyro_error(self, gyro_error='low-accuracy'):      # imu_model.py:207-236
        self.gyro_err = self._set(self.gyro_err, gyro_error, 0, 'gyro_error')

    def set_accel_error(self, accel_error='low-accuracy'):    # imu_model.py:238-267
        self.accel_err = self._set(self.accel_err, accel_error, 1, 'accel_error')

    def set_mag_error(self, mag_error='low-accuracy'):        # imu_model.py:321-352
        if self.magnetometer:
            self.mag_err = self._set(self.mag_err, mag_error, 2, 'mag_error')

    def set_gps(self, gps_error=None):                         # imu_model.py:269-290
        if self.gps:
            self.gps_err = self._opt(gps_error, gps_low_accuracy, ('stdp', 'stdv'), 'gps_error')

    def set_odo(self, odo_error=None):                         # imu_model.py:292-319
        if self.odo:
            self.odo_err = self._opt(odo_error, odo_low_accuracy, ('scale', 'stdv'), 'odo_error')
