"""C3-shaped timing (long_drive, rf 0, stats-only) on a shortened horizon -- development aid."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0)
t = {k: (v[:20000] if hasattr(v, 'shape') and v.shape and v.shape[0] > 20000 else v) for k, v in truth.items()}
for prec, kw in (('f64', {}), ('f64', {'proc_first': 0, 'end_ned': True}), ('f32', {})):
    job = ginsim.MonteCarloJob(ctx, 200.0, 0, t, acc, gyr, ini, runs=262144, seed=1, precision=prec, **kw).run()
    ts = []
    for _ in range(3):
        ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
    print(prec, job.kernel_name(), 'n=20000 R=262144: min %.2f ms -> %.3g sample*MC/s' % (min(ts), 20000 * 262144 / min(ts) * 1e3))
    job.release()
