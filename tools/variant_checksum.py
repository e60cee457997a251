"""Checksum of one small batch through whatever kernel variant GINSIM_SPLIT selects (development aid)."""
import os, sys, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
h = hashlib.sha256()
for prec in ('f64', 'f32'):
    for rf, algos in ((1, ('free',)), (0, ('free', 'odo'))):
        ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
        t = {k: (v[:61] if hasattr(v, 'shape') and v.shape and v.shape[0] == 1000 else v) for k, v in truth.items()}
        job = ginsim.MonteCarloJob(ctx, 100.0, rf, t, acc, gyr, ini, runs=1000, algos=algos, odo_err={'scale': 0.999, 'stdv': 0.1},
                                   seed=5, keep_sensors=True, keep_traj=True, precision=prec).run()
        for a in algos:
            h.update(np.ascontiguousarray(job.end_errors(a)).tobytes())
            for x in job.trajectories(a, [0, 63, 64, 999]):
                h.update(np.ascontiguousarray(x).tobytes())
        h.update(np.ascontiguousarray(job.sensors('accel', [1, 998])).tobytes())
        h.update(np.ascontiguousarray(job.sensors('gyro', [1, 998])).tobytes())
        print(prec, rf, algos, job.kernel_name(), h.hexdigest()[:16])
        job.release()
