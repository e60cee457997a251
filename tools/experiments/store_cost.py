"""Kernel time against what is materialised (development aid): none / sensors (6 planes) / trajectories (9) / all (15)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
P = os.environ.get('AB_PREC', 'f64')
for R in (65536, 262144):
    res = []
    for ks, kt in ((False, False), (True, False), (False, True), (True, True)):
        ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
        job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=ks, keep_traj=kt, precision=P)
        job.run()
        ts = []
        for _ in range(15):
            ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
        res.append('sens=%d traj=%d: min %.3f med %.3f' % (ks, kt, min(ts), np.median(ts)))
        job.release()
    print(P, R, job.kernel_name(), ' | '.join(res))
