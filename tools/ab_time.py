"""Timing of mc_kernel for A/B comparisons (development aid): prints min/median over many launches."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
res = []
P = os.environ.get('AB_PREC', 'f64')
for rf, R, keep, prec in ((1, 65536, True, P), (1, 65536, False, P), (1, 262144, True, P), (1, 262144, False, P), (0, 65536, True, P)):
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=keep, keep_traj=keep, precision=prec)
    job.run()
    ts = []
    for _ in range(12):
        ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
    res.append('rf%d R=%d keep=%d: min %.3f med %.3f' % (rf, R, keep, min(ts), np.median(ts)))
    job.release()
print(os.environ.get('GINSIM_LIB', 'default').split('/')[-1], ' | '.join(res))
