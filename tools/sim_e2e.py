"""End-to-end wall time of the drop-in Sim for BASELINE config 2 (development aid)."""
import os, sys, time, io, contextlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
t0 = time.perf_counter()
from gnss_ins_sim.sim import imu_model, ins_sim
from demo_algorithms import free_integration
t_import = time.perf_counter() - t0
csv = os.path.join(REPO, 'gnss-ins-sim_amd', 'motion_profiles', 'turn_90deg.csv')
ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
ini[0:2] *= np.pi / 180; ini[6:9] *= np.pi / 180
for R, keep in ((1000, True), (65536, True), (65536, False), (262144, False)):
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    algo = free_integration.FreeIntegration(ini)
    t0 = time.perf_counter()
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, mode=None, env=None, algorithm=algo, seed=1,
                      keep_trajectories=keep)
    t1 = time.perf_counter()
    sim.run(R)
    t2 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results('', err_stats_start=-1)
    t3 = time.perf_counter()
    x = sim.dmgr.pos.data['algo0_5'] if keep else None
    t4 = time.perf_counter()
    print('R=%d keep=%s: ctor %.3f s, run %.3f s, results %.3f s, one-run view %.4f s  -> %.3g sample*MC/s end to end'
          % (R, keep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, R * 1000 / (t3 - t0)))
    del sim
print('import %.2f s' % t_import)
