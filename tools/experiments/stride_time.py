"""Does the power-of-two plane stride of the C2 shape (65 536 runs x 1000 samples x 8 B = 2^19 x 1000 B) cost anything?  Times the
materialising fp64 / fp32 launch at 65 536 runs and at run counts whose planes are not multiples of a large power of two
(development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
for P in ('f64', 'f32'):
    for R in (65536, 65536 - 256, 65536 - 64, 65536, 65536 - 4096 + 64):
        job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=True, keep_traj=True, precision=P)
        job.run()
        spent = 0.0
        while spent < 40.0:
            ctx.timer_begin(); job.launch(); spent += ctx.timer_end()
        for i in range(20):
            ctx.event_record(2 * i); job.launch(); ctx.event_record(2 * i + 1)
        ts = [ctx.event_elapsed(2 * i, 2 * i + 1) for i in range(20)]
        b = (120 if P == 'f64' else 60) * R * 1000
        print('%s R=%6d: avg %.3f ms  %.0f GB/s  %.3f of 8 TB/s   (%.4f ns per run-step)' % (P, R, np.mean(ts), b / np.mean(ts) / 1e6, b / np.mean(ts) / 8e9, np.mean(ts) * 1e6 / (R * 1000) * 1e0))
        job.release()
