#!/usr/bin/env python3
"""Allan deviation of simulated IMU noise, entirely on the device: static profile, R runs, the Allan plugin.

    PYTHONPATH=gnss-ins-sim_amd python examples/demo_allan.py [runs]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd'))

from gnss_ins_sim.sim import imu_model, ins_sim      # noqa: E402
from demo_algorithms import allan_analysis           # noqa: E402

MOTION = os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd', 'motion_profiles', 'static_1800s.csv')


def main(runs):
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    sim = ins_sim.Sim([400.0, 0.0, 0.0], MOTION, ref_frame=1, imu=imu, mode=None, env=None, algorithm=allan_analysis.Allan())
    t0 = time.perf_counter()
    sim.run(runs)
    dt = time.perf_counter() - t0
    tau = sim.dmgr.algo_time.data['algo0_0']
    ad = np.stack([sim.dmgr.ad_gyro.data['algo0_%d' % r] for r in range(runs)])        # (runs, ntau, 3)
    arw = 0.25 / 60.0 * np.pi / 180.0                                                  # rad/s/sqrt(Hz)
    k = int(np.argmin(np.abs(tau - 1.0)))
    print('%d runs x %d samples in %.3f s; %d averaging times' % (runs, sim.dmgr.time.data.shape[0], dt, tau.size))
    print('gyro AD at tau = 1 s: %.3e +- %.1e rad/s (model ARW %.3e)' % (ad[:, k, :].mean(), ad[:, k, :].std(), arw))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
