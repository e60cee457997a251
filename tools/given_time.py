"""Timing of the given-sensors mechanisation kernel (reads 48 B, writes 72 B per sample*MC) -- development aid."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
for rf in (1, 0):
    for R in (65536, 131072):
        ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
        n = truth['ref_accel'].shape[0]
        gen = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=True, keep_traj=False).run()
        given = {'gyro': gen.buffer('gyro'), 'accel': gen.buffer('accel')}
        for keep in (True, False):
            rep = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, None, None, ini, runs=R, keep_traj=keep, given=given).run()
            ts = []
            for _ in range(10):
                ctx.timer_begin(); rep.launch(); ts.append(ctx.timer_end())
            b = (48 + (72 if keep else 0)) * R * n
            print('rf%d R=%d keep_traj=%d %s: min %.3f med %.3f ms  %.0f GB/s (min)' % (rf, R, keep, rep.kernel_name(), min(ts), np.median(ts), b / min(ts) / 1e6))
            rep.release()
        gen.release()
