"""Timing of the fp32 (and fp64) MC kernels at the bench shapes for A/B comparisons (development aid).  Every shape is warmed
up by time (40 ms) and then timed over 20 back-to-back launches (HIP events)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
P = os.environ.get('AB_PREC', 'f32')
shapes = [(1, 65536, True), (1, 65536, False), (1, 262144, True), (0, 65536, True)]
if os.environ.get('AB_SHAPES') == 'c4':
    shapes = [(1, 65536, True), (1, 131072, True), (1, 262144, True), (0, 65536, False)]
if os.environ.get('AB_SHAPES') == 'c2':
    shapes = shapes[:2]
out = []
for rf, R, keep in shapes:
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=keep, keep_traj=keep, precision=P)
    if os.environ.get('AB_PLAIN'):
        job.params.block_threads = 256
    job.run()
    spent = 0.0
    while spent < 40.0:
        ctx.timer_begin(); job.launch(); spent += ctx.timer_end()
    reps = 20
    for i in range(reps):
        ctx.event_record(2 * i); job.launch(); ctx.event_record(2 * i + 1)
    ts = [ctx.event_elapsed(2 * i, 2 * i + 1) for i in range(reps)]
    b = (60 if P == 'f32' else 120) * R * 1000 if keep else 0
    out.append('rf%d R=%d keep=%d %s: avg %.3f min %.3f ms%s' % (rf, R, keep, job.kernel_name().split('::')[-1], np.mean(ts), min(ts),
               '  %.0f GB/s %.3f' % (b / np.mean(ts) / 1e6, b / np.mean(ts) / 1e6 / 8000) if keep else ''))
    job.release()
print(P, 'PROD=' + os.environ.get('GINSIM_SPLIT_PROD', '-'), 'plain' if os.environ.get('AB_PLAIN') else '', '\n   ' + '\n   '.join(out))
