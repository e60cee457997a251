import os, sys, time
sys.path[:0] = ['/root/repo/gnss-ins-sim_amd', '/root/repo']
import numpy as np
import ginsim
from ginsim import workloads
from ginsim._lib import check
lib = ginsim.lib
ctx = ginsim.Context(0)
text = open(workloads.profile_path('static_1800s')).read().split('\n')
ini, _ = workloads.parse_motion('\n'.join(text[:4]))
seg = np.array([[1.0, 0, 0, 0, 0, 0, 0, 3600.0, 0.0]])
raw = ginsim.pathgen(ini, seg, 400.0, 0.0, workloads.HIGH_MOBILITY, 1)
truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
         'ref_pos': raw['nav'][:, 1:4], 'ref_vel': raw['nav'][:, 4:7], 'ref_att': raw['nav'][:, 7:10]}
acc, gyr = workloads.imu_grade('mid-accuracy')
job = ginsim.MonteCarloJob(ctx, 400.0, 1, truth, acc, gyr, None, runs=32, algos=(), seed=1, keep_sensors=True)
job.run()
for rep in range(4):
    t0 = time.perf_counter(); tau, ad = job.allan(400.0); t1 = time.perf_counter()
    print('job.allan wall %.3f ms' % ((t1 - t0) * 1e3))
n, R = job.n, job.runs
per = 3 * n * R * 8
for rep in range(3):
    t0 = time.perf_counter(); tmp = ctx.malloc(2 * per); t1 = time.perf_counter()
    for i, nm in enumerate(('accel', 'gyro')):
        check(lib.ginsim_runs_to_series(ctx.handle, job.buffer(nm).ptr, 3, n, R, tmp.at(i * per)))
    ctx.synchronize() if hasattr(ctx, 'synchronize') else None
    t2 = time.perf_counter()
    av, tau = ginsim.allan_var(ctx, tmp, n, 6 * R, n, 400.0); t3 = time.perf_counter()
    tmp.free(); t4 = time.perf_counter()
    print('malloc %.3f relayout(+sync?) %.3f allan %.3f free %.3f ms' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
