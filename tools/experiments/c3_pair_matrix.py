#!/usr/bin/env python3
"""Two launches of BASELINE config 3's shape at once -- 256 runs with everything kept (ONE workgroup) and 261 888 runs statistics
only -- on every ordered pair of four contexts of one process, after a PSD vibration job (hipFFT) ran or not: wall of the pair,
each kernel's own duration, and the XCD on which each context's launches start (ginsim_stream_first_xcc): the pair is free when
the block's workgroup lands on the die the statistics launch leaves a slot on, (first(rest) + 1023) mod 8.
    python tools/experiments/c3_pair_matrix.py [psd_first]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gnss-ins-sim_amd'))
import ginsim                                           # noqa: E402
from ginsim import workloads                            # noqa: E402

ctx0 = ginsim.default_context()
if 'psd_first' in sys.argv:
    f = np.array([0.0, 8.0, 11.0, 13.0, 16.0, 50.0])
    v = {'type': 'psd', 'freq': f, 'x': np.full(6, 1e-3), 'y': np.full(6, 1e-3), 'z': np.full(6, 2e-3)}
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    ginsim.MonteCarloJob(ctx0, 100.0, 1, truth, acc, gyr, ini, runs=4096, seed=3, keep_sensors=True, vib_accel=v).run().release()
    print('a PSD vibration job ran')
ctxs = [ctx0] + [ginsim.Context(0) for _ in range(3)]
print('first xcc of the four contexts, asked three times:', [[c.first_xcc() for c in ctxs] for _ in range(3)], flush=True)
ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0, fs_gps=10.0, gps=True)
acc, gyr = workloads.imu_grade('mid-accuracy')
jobs = {}
for i, c in enumerate(ctxs):
    blk = ginsim.MonteCarloJob(c, 200.0, 0, truth, acc, gyr, ini, runs=256, seed=7, keep_sensors=True, keep_traj=True, proc_first=0, end_ned=True)
    rest = ginsim.MonteCarloJob(c, 200.0, 0, truth, acc, gyr, ini, runs=262144 - 256, run_offset=256, seed=7, proc_first=0, end_ned=True)
    blk.run()
    jobs[i] = (blk, rest)
for i in (0, 1):
    c = ctxs[i]
    for nm, j in (('block', jobs[i][0]), ('rest', jobs[i][1])):
        c.event_record(0)
        j.launch()
        c.event_record(1)
        print('ctx %d alone: %-5s %.1f ms' % (i, nm, c.event_elapsed(0, 1)), flush=True)
for a in range(4):
    for b in range(4):
        if a == b:
            continue
        ca, cb = ctxs[a], ctxs[b]
        ca.sync(); cb.sync()
        t0 = time.perf_counter()
        ca.event_record(0); jobs[a][0].launch(); ca.event_record(1)
        cb.event_record(0); jobs[b][1].launch(); cb.event_record(1)
        ca.sync(); cb.sync()
        wall = (time.perf_counter() - t0) * 1e3
        xa, xb = ca.first_xcc(), cb.first_xcc()
        print('block on ctx %d (xcc %d), rest on ctx %d (xcc %d): wall %.0f ms  block %.0f ms  rest %.0f ms   predicted free: %s' % (
            a, xa, b, xb, wall, ca.event_elapsed(0, 1), cb.event_elapsed(0, 1), xa == (xb + 1023) % 8), flush=True)
