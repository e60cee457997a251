#!/usr/bin/env python3
"""Does the one-workgroup launch that materialises a few kept runs of C3 run underneath the statistics launch over all runs
(Sim(keep_runs=2) on long_drive @200 Hz, 262 144 runs)?  Times the pieces."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads

ctx = ginsim.default_context()
side = ginsim.Context(0)
fs, rf = 200.0, 0
ini, truth, _ = workloads.truth_from_profile('long_drive', fs, rf)
acc, gyr = workloads.imu_grade('mid-accuracy')
R = int(os.environ.get('R', 262144))
for rep in range(2):
    kept = ginsim.MonteCarloJob(side, fs, rf, truth, acc, gyr, ini, runs=2, seed=5, keep_sensors=True, keep_traj=True)
    t0 = time.perf_counter(); kept.launch(); side.sync(); t_k = time.perf_counter() - t0
    big = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, seed=5, proc_first=0, end_ned=True)
    t0 = time.perf_counter(); big.launch(); ctx.sync(); t_b = time.perf_counter() - t0
    t0 = time.perf_counter(); kept.launch(); big.launch(); ctx.sync(); side.sync(); t_both = time.perf_counter() - t0
    t0 = time.perf_counter(); big.launch(); kept.launch(); ctx.sync(); side.sync(); t_both2 = time.perf_counter() - t0
    print('rep %d: kept alone %.3f s (%s), all runs alone %.3f s, kept then all %.3f s, all then kept %.3f s' % (rep, t_k, kept.kernel_name(), t_b, t_both, t_both2), flush=True)
    kept.release(); big.release()
