#!/usr/bin/env python3
"""One C3 launch (long_drive @200 Hz x RUNS runs, ref_frame 0, free integration, online process statistics), timed: for A/B
of the dispatch (GINSIM_PS_SIMPLE etc.).  Prints kernel name, seconds of two launches, a checksum of the statistics."""
import hashlib, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads

ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0)
cut = int(os.environ.get('SAMPLES', 0))
if cut:
    truth = {k: (v[:cut] if hasattr(v, 'shape') and v.shape and v.shape[0] >= cut else v) for k, v in truth.items()}
acc, gyr = workloads.imu_grade('mid-accuracy')
R = int(os.environ.get('RUNS', 262144))
job = ginsim.MonteCarloJob(ctx, 200.0, 0, truth, acc, gyr, ini, runs=R, seed=11, proc_first=0, end_ned=True)
ts = []
for _ in range(2):
    ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end() * 1e-3)
ps = job.process_stats_online('free')
print(json.dumps({'kernel': job.kernel_name(), 'runs': R, 'n': job.n, 's': ts, 'sample_MC_per_s': R * job.n / min(ts),
                  'stats_sha': hashlib.sha256(np.ascontiguousarray(ps).tobytes()).hexdigest()[:16],
                  'end_sha': hashlib.sha256(job.end_errors('free').tobytes()).hexdigest()[:16]}))
