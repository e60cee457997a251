#!/usr/bin/env python3
"""Where the 0.15 s of Sim.results() on BASELINE config 3 go (262 144 runs, per-run process statistics): cProfile, top entries."""
import contextlib
import cProfile
import io
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gnss-ins-sim_amd'))
from ginsim import workloads                            # noqa: E402
from gnss_ins_sim.sim import imu_model, ins_sim         # noqa: E402
from demo_algorithms import free_integration            # noqa: E402

csv = workloads.profile_path('long_drive')
ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
ini[0:2] *= np.pi / 180
ini[6:9] *= np.pi / 180
for rep in range(2):
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=9, gps=True)
    sim = ins_sim.Sim([200.0, 10.0, 200.0], csv, ref_frame=0, imu=imu, mode=None, env=None,
                      algorithm=free_integration.FreeIntegration(ini), seed=7, geo_mag_n=[33.0, -2.4, 36.5], keep_runs=2)
    sim.run(262144)
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable()
        sim.results(err_stats_start=0)
        pr.disable()
    if rep == 1:
        st = pstats.Stats(pr)
        st.sort_stats('cumulative').print_stats(18)
