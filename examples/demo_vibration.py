#!/usr/bin/env python3
"""Free integration in a vibration environment on an MI355X: the reference's Sim(env=...) strings (ins_sim.py:108-124) through the
drop-in package -- the same Monte Carlo four times (no vibration, random, sinusoidal, a power spectral density as an (n, 4) array) and
what the environment does to the end-point attitude and velocity errors.

    PYTHONPATH=gnss-ins-sim_amd python examples/demo_vibration.py [runs]
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd'))

from gnss_ins_sim.sim import imu_model, ins_sim        # noqa: E402
from demo_algorithms import free_integration           # noqa: E402

D2R = np.pi / 180.0
MOTION = os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd', 'motion_profiles', 'turn_90deg.csv')
ENVS = [('no vibration', None),
        ('random: 0.03 g / 0.5 deg/s rms', {'acc': '[0.03 0.03 0.03]g-random', 'gyro': '[0.5 0.5 0.5]d-random'}),
        ('sinusoidal: 0.05 g at 25 Hz, 0.3 deg/s at 2 Hz', {'acc': '[0.05 0.05 0.05]g-25Hz-sinusoidal', 'gyro': '[0.3 0.3 0.3]d-2Hz-sinusoidal'}),
        # single-sided PSDs, rows [freq, x, y, z]: (m/s^2)^2/Hz and (rad/s)^2/Hz -- a resonance at 12 Hz on a floor (ins_sim.py:115-121)
        ('psd: 12 Hz resonance, 0.3 m/s^2 / 0.4 deg/s rms', {
            'acc': np.array([[f, p, p, p] for f, p in ((0.0, 1e-4), (8.0, 1e-4), (11.0, 2e-2), (13.0, 2e-2), (16.0, 1e-4), (50.0, 1e-4))]),
            'gyro': np.array([[f, p, p, p] for f, p in ((0.0, 1e-8), (8.0, 1e-8), (11.0, 1e-5), (13.0, 1e-5), (16.0, 1e-8), (50.0, 1e-8))])})]


def main(runs):
    ini = np.genfromtxt(MOTION, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= D2R
    ini[6:9] *= D2R
    for label, env in ENVS:
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
        sim = ins_sim.Sim([100.0, 0.0, 0.0], MOTION, ref_frame=1, imu=imu, mode=None, env=env,
                          algorithm=free_integration.FreeIntegration(ini), seed=2024)
        t0 = time.perf_counter()
        sim.run(runs)
        dt = time.perf_counter() - t0
        with contextlib.redirect_stdout(io.StringIO()):
            sim.results(err_stats_start=-1)
        att, vel = sim.err_stats['att_euler'], sim.err_stats['vel']
        print('%-48s %d runs in %.1f ms   att std [deg] %s   vel std [m/s] %s' % (
            label, runs, dt * 1e3, np.array2string(np.asarray(att['std']), precision=4), np.array2string(np.asarray(vel['std']), precision=4)))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
