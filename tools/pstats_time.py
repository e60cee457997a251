"""Timing of the process-error statistics (err_stats_start >= 0) on kept trajectories -- development aid."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
for rf in (1, 0):
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=65536, seed=1, keep_traj=True).run()
    for ned in ((False, True) if rf == 0 else (False,)):
        job.process_stats('free', 0, ned)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.timer_begin(); out = job.process_stats('free', 0, ned); ms = ctx.timer_end(); ts.append((ms, (time.perf_counter() - t0) * 1e3))
        print('rf%d ned=%d: device %.3f ms, wall %.3f ms  (4.72 GB read -> %.0f GB/s device)' % (rf, ned, min(t[0] for t in ts), min(t[1] for t in ts), 4718.6 / min(t[0] for t in ts)))
    job.release()
