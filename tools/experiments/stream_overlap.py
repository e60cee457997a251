#!/usr/bin/env python3
"""Do kernels of two contexts (two HIP streams) of one process overlap on the GPU box?  Pairs of long one-workgroup launches
(2 runs x long_drive @200 Hz: a sequential chain of 193 036 steps) and a one-workgroup launch next to a chip-filling one."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import ginsim
from ginsim import workloads

a, b = ginsim.Context(0), ginsim.Context(0)
fs, rf = 200.0, 0
ini, truth, _ = workloads.truth_from_profile('long_drive', fs, rf)
acc, gyr = workloads.imu_grade('mid-accuracy')
mk = lambda c, runs, **kw: ginsim.MonteCarloJob(c, fs, rf, truth, acc, gyr, ini, runs=runs, seed=5, **kw)
ja, jb = mk(a, 2, keep_sensors=True, keep_traj=True), mk(b, 2, keep_sensors=True, keep_traj=True)
for j in (ja, jb):
    j.run()
t0 = time.perf_counter(); ja.launch(); a.sync(); one = time.perf_counter() - t0
t0 = time.perf_counter(); ja.launch(); jb.launch(); a.sync(); b.sync(); two = time.perf_counter() - t0
print('one-workgroup launch alone %.3f s; two of them on two streams %.3f s' % (one, two), flush=True)
for runs in (4096, 65536, 262144):
    big = mk(b, runs)            # end-point statistics only: the wave-specialised kernel, one 768-thread workgroup per 256 runs
    big.run()
    t0 = time.perf_counter(); big.launch(); b.sync(); alone = time.perf_counter() - t0
    t0 = time.perf_counter(); ja.launch(); big.launch(); a.sync(); b.sync(); both = time.perf_counter() - t0
    t0 = time.perf_counter(); big.launch(); ja.launch(); a.sync(); b.sync(); both2 = time.perf_counter() - t0
    print('%6d runs (%s): alone %.3f s, small first %.3f s, big first %.3f s' % (runs, big.kernel_name(), alone, both, both2), flush=True)
    big.release()
