#!/usr/bin/env python3
"""The drop-in Sim on BASELINE config 3 as named (262 144 runs, statistics only, two kept runs as ONE workgroup on a sibling
context next to the statistics launch): wall of run() before and after something else in the process has created streams of its
own (here: the PSD vibration, whose hipFFT plans do) -- with the sibling's stream at the highest priority (default) and without
($GINSIM_SIBLING_PROBE=0: round 5's rule, the block on the older of the two streams).
    python tools/experiments/c3_sim_queues.py [psd_first]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gnss-ins-sim_amd'))
import ginsim                                           # noqa: E402
from ginsim import workloads                            # noqa: E402
from gnss_ins_sim.sim import imu_model, ins_sim         # noqa: E402
from demo_algorithms import free_integration            # noqa: E402


def c3(tag):
    csv = workloads.profile_path('long_drive')
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    walls = []
    for rep in range(3):
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=9, gps=True)
        t0 = time.perf_counter()
        sim = ins_sim.Sim([200.0, 10.0, 200.0], csv, ref_frame=0, imu=imu, mode=None, env=None,
                          algorithm=free_integration.FreeIntegration(ini), seed=7, geo_mag_n=[33.0, -2.4, 36.5], keep_runs=2)
        sim.run(262144)
        walls.append(time.perf_counter() - t0)
        del sim
    hit = ins_sim.Sim._SIBLINGS.get(0)
    main = ginsim.default_context()
    where = 'block on xcc %d, statistics on xcc %d (W mod 8 = %d)' % (hit['pair'][0].first_xcc(), hit['pair'][1].first_xcc(), hit['want']) if hit else ''
    print('%-28s C3 Sim.run walls %s s   %s; default context xcc %d, is rest: %s' % (tag, ' '.join('%.3f' % w for w in walls), where, main.first_xcc(), hit and hit['pair'][1] is main), flush=True)


def psd():
    f = np.array([0.0, 8.0, 11.0, 13.0, 16.0, 50.0])
    v = {'type': 'psd', 'freq': f, 'x': np.full(6, 1e-3), 'y': np.full(6, 1e-3), 'z': np.full(6, 2e-3)}
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ginsim.default_context(), 100.0, 1, truth, acc, gyr, ini, runs=4096, seed=3, keep_sensors=True, vib_accel=v).run()
    job.release()
    print('a PSD vibration job ran (hipFFT loaded, plans made)', flush=True)


if __name__ == '__main__':
    if 'psd_first' in sys.argv:
        psd()
        c3('after the PSD job')
    else:
        c3('fresh process')
        psd()
        c3('after the PSD job')
