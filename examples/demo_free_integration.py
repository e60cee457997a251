#!/usr/bin/env python3
"""Free-integration Monte Carlo on an MI355X through the drop-in package: the calling sequence of the reference's
demo_free_integration.py (Sim + IMU + two plugins + run + results), with 65 536 runs instead of 10.

    PYTHONPATH=gnss-ins-sim_amd python examples/demo_free_integration.py [runs]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd'))

from gnss_ins_sim.sim import imu_model, ins_sim                      # noqa: E402
from demo_algorithms import free_integration, free_integration_odo   # noqa: E402

D2R = np.pi / 180.0
MOTION = os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd', 'motion_profiles', 'turn_90deg.csv')


def main(runs):
    imu_err = {'gyro_b': np.zeros(3), 'gyro_arw': np.full(3, 0.25), 'gyro_b_stability': np.full(3, 3.5),
               'gyro_b_corr': np.full(3, 100.0),
               'accel_b': np.zeros(3), 'accel_vrw': np.array([0.03119, 0.03009, 0.04779]),
               'accel_b_stability': np.array([4.29e-5, 5.72e-5, 8.02e-5]), 'accel_b_corr': np.full(3, 200.0)}
    imu = imu_model.IMU(accuracy=imu_err, axis=6, gps=False, odo=True, odo_opt={'scale': 0.999, 'stdv': 0.1})
    ini = np.genfromtxt(MOTION, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= D2R
    ini[6:9] *= D2R
    algos = [free_integration_odo.FreeIntegration(ini), free_integration.FreeIntegration(ini)]
    sim = ins_sim.Sim([100.0, 0.0, 0.0], MOTION, ref_frame=1, imu=imu, mode=None, env=None, algorithm=algos)
    t0 = time.perf_counter()
    sim.run(runs)
    t1 = time.perf_counter()
    sim.results(err_stats_start=-1)                 # end-point statistics over all runs, printed like the reference
    n = sim.dmgr.time.data.shape[0]
    print('%d runs x %d samples x %d algorithms in %.3f s' % (runs, n, len(algos), t1 - t0))
    one = sim.dmgr.pos.data.get('algo1_7') if hasattr(sim.dmgr.pos.data, 'get') else None
    if one is not None:
        print('run 7, free integration, final position (virtual inertial frame, m):', one[-1])


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
