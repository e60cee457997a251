"""A/B timing of one library build (GINSIM_LIB=<path>) on the launches an instruction-stream experiment is judged on:
a C3-shaped statistics-only launch (long_drive @200 Hz cut to 8192 samples, 262 144 runs, with and without the online process
statistics), the fp32 and fp64 C2 launches materialised and with nothing kept.  Prints kernel times (min / avg of `reps`
back-to-back launches after a warm-up by time) and a checksum of the end-point errors, so that two builds can be compared for
speed AND for bit-identity.  Development aid:  GINSIM_LIB=lib/libginsim_<tag>.so python tools/experiments/ab_kernels.py"""
import hashlib
import json
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
import bench

ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
out = {'lib': os.path.basename(ginsim.LIB_PATH)}
reps = int(os.environ.get('REPS', 8))


def leg(tag, profile, fs, rf, R, keep, precision, cut=None, **kw):
    ini, truth, _ = workloads.truth_from_profile(profile, fs, rf)
    if cut:
        truth = bench.cut_truth(truth, cut)
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, seed=7, keep_sensors=keep, keep_traj=keep, precision=precision, **kw)
    job.run()
    avg, mn = bench.time_launches(ctx, job.launch, reps)
    e = job.end_errors('free')
    out[tag] = {'ms_avg': round(avg, 4), 'ms_min': round(mn, 4), 'kernel': job.kernel_name(),
                'end_sha': hashlib.sha256(np.ascontiguousarray(e).tobytes()).hexdigest()[:12]}
    job.release()


leg('c3_ps', 'long_drive', 200.0, 0, 262144, False, 'f64', cut=8192, proc_first=0, end_ned=True)
leg('c3_end', 'long_drive', 200.0, 0, 262144, False, 'f64', cut=8192)
leg('f32_nothing', 'turn_90deg', 100.0, 1, 65536, False, 'f32')
leg('f32_kept', 'turn_90deg', 100.0, 1, 65536, True, 'f32')
leg('f64_nothing', 'turn_90deg', 100.0, 1, 65536, False, 'f64')
leg('f64_kept', 'turn_90deg', 100.0, 1, 65536, True, 'f64')
print(json.dumps(out))
