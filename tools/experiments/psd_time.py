#!/usr/bin/env python3
"""What the PSD vibration costs on BASELINE config 2's shape (65 536 runs x 1000 samples, fp64, everything kept): the series of both
sensors (ginsim_vib_psd_series: spectrum kernel, batched inverse FFT, transposition) and the launch that reads them.
    python tools/experiments/psd_time.py            (or under rocprofv3 --kernel-trace --stats)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gnss-ins-sim_amd'))
import ginsim                               # noqa: E402
from ginsim import workloads                # noqa: E402

ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
acc, gyr = workloads.imu_grade('mid-accuracy')
f = np.array([0.0, 8.0, 11.0, 13.0, 16.0, 50.0])
psd = lambda u: {'type': 'psd', 'freq': f, 'x': u * np.array([1e-4, 1e-4, 2e-2, 2e-2, 1e-4, 1e-4]), 'y': u * np.full(6, 1e-3), 'z': u * np.full(6, 2e-3)}
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for rep in range(3):
    ctx.sync()
    t0 = time.perf_counter()
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=3, keep_sensors=True, keep_traj=True, vib_accel=psd(1.0), vib_gyro=psd(1e-4))
    ctx.sync()
    t1 = time.perf_counter()
    job.run()
    ms = []
    for i in range(5):
        ctx.event_record(0)
        job.launch()
        ctx.event_record(1)
        ms.append(ctx.event_elapsed(0, 1))
    plain = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=3, keep_sensors=True, keep_traj=True).run()
    ctx.event_record(0)
    plain.launch()
    ctx.event_record(1)
    print('rep %d: job with both series made in %.1f ms; launch %s ms (%s); without vibration %.3f ms (%s)' % (
        rep, (t1 - t0) * 1e3, ' '.join('%.3f' % m for m in ms), job.kernel_name(), ctx.event_elapsed(0, 1), plain.kernel_name()), flush=True)
    job.release()
    plain.release()
