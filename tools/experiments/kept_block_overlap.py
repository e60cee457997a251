#!/usr/bin/env python3
"""C3 as Sim runs it with keep_runs: can the kept runs ride along as the FIRST WORKGROUP of the batch?  Kept job = runs 0..255 (one
256-run workgroup, everything materialised, online statistics) on one context, the statistics-only job = runs 256..262143 (1023
workgroups) on another: 1024 workgroups in all, as many as the one launch that integrates every run."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import ginsim
from ginsim import workloads

a, b = ginsim.Context(0), ginsim.Context(0)
fs, rf, R = 200.0, 0, 262144
ini, truth, _ = workloads.truth_from_profile('long_drive', fs, rf, fs_gps=10.0, gps=True)
acc, gyr = workloads.imu_grade('mid-accuracy')
mk = lambda c, runs, off=0, **kw: ginsim.MonteCarloJob(c, fs, rf, truth, acc, gyr, ini, runs=runs, seed=5, run_offset=off, ini_first=off, **kw)
ps = dict(proc_first=0, end_ned=True)
whole = mk(b, R, **ps)
whole.run()
t0 = time.perf_counter(); whole.launch(); b.sync(); t_whole = time.perf_counter() - t0
two = mk(a, 2, keep_sensors=True, keep_traj=True)
two.run()
t0 = time.perf_counter(); two.launch(); a.sync(); t_two = time.perf_counter() - t0
print('all %d runs, statistics only: %.3f s (%s); two kept runs alone: %.3f s (%s)' % (R, t_whole, whole.kernel_name(), t_two, two.kernel_name()), flush=True)
whole.release()
for kb in (256, 512):
    try:
        kept = mk(a, kb, keep_sensors=True, keep_traj=True, **ps)
    except Exception as e:
        print('kept block with online statistics refused: %r' % (e,)); kept = mk(a, kb, keep_sensors=True, keep_traj=True)
    kept.run()
    t0 = time.perf_counter(); kept.launch(); a.sync(); t_kept = time.perf_counter() - t0
    rest = mk(b, R - kb, off=kb, **ps)
    rest.run()
    t0 = time.perf_counter(); rest.launch(); b.sync(); t_rest = time.perf_counter() - t0
    t0 = time.perf_counter(); kept.launch(); rest.launch(); a.sync(); b.sync(); t_both = time.perf_counter() - t0
    t0 = time.perf_counter(); rest.launch(); kept.launch(); a.sync(); b.sync(); t_both2 = time.perf_counter() - t0
    print('kept block of %d runs (%s): alone %.3f s; the other %d runs alone %.3f s; kept first %.3f s; kept second %.3f s' %
          (kb, kept.kernel_name(), t_kept, R - kb, t_rest, t_both, t_both2), flush=True)
    kept.release(); rest.release()
