#!/usr/bin/env python3
"""kept_block_overlap.py gave block || rest = 0.990 s; inside Sim the same two launches take 1.30 s.  What differs?  The order in
which the two contexts were created, and whether the jobs had been launched before."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import ginsim
from ginsim import workloads

order = os.environ.get('ORDER', 'rest_first')
if order == 'rest_first':
    b = ginsim.Context(0); a = ginsim.Context(0)
else:
    a = ginsim.Context(0); b = ginsim.Context(0)
fs, rf, R, kb = 200.0, 0, 262144, 256
ini, truth, _ = workloads.truth_from_profile('long_drive', fs, rf, fs_gps=10.0, gps=True)
acc, gyr = workloads.imu_grade('mid-accuracy')
mk = lambda c, runs, off=0, **kw: ginsim.MonteCarloJob(c, fs, rf, truth, acc, gyr, ini, runs=runs, seed=5, run_offset=off, ini_first=off, **kw)
ps = dict(proc_first=0, end_ned=True)
for rep in range(3):
    kept = mk(a, kb, keep_sensors=True, keep_traj=True, **ps)
    rest = mk(b, R - kb, off=kb, **ps)
    t0 = time.perf_counter(); kept.launch(); rest.launch(); b.sync(); t_rest = time.perf_counter() - t0; a.sync(); t_both = time.perf_counter() - t0
    t0 = time.perf_counter(); kept.launch(); rest.launch(); a.sync(); t_kept = time.perf_counter() - t0; b.sync(); t_both2 = time.perf_counter() - t0
    print('%s rep %d: fresh jobs: rest done %.3f, both %.3f s; again: kept done %.3f, both %.3f s' % (order, rep, t_rest, t_both, t_kept, t_both2), flush=True)
    kept.release(); rest.release()
