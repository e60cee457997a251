"""End-to-end demo_allan flow (BASELINE config 5, second half): static profile, Sim.run(1) with the Allan plugin -- development aid."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
from gnss_ins_sim.sim import imu_model, ins_sim
from demo_algorithms import allan_analysis
csv = os.path.join(REPO, 'gnss-ins-sim_amd', 'motion_profiles', 'static_1800s.csv')
for fs, R in ((100.0, 1), (400.0, 1), (400.0, 32), (400.0, 32)):
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    algo = allan_analysis.Allan()
    t0 = time.perf_counter()
    sim = ins_sim.Sim([fs, 0.0, 0.0], csv, ref_frame=1, imu=imu, mode=None, env=None, algorithm=algo, seed=3)
    t1 = time.perf_counter()
    sim.run(R)
    t2 = time.perf_counter()
    ad = sim.dmgr.ad_gyro.data
    t3 = time.perf_counter()
    n = sim.dmgr.time.data.shape[0]
    print('fs=%g n=%d R=%d: ctor %.3f s, run %.3f s, fetch %.3f s  (%.3g samples*run/s)' % (fs, n, R, t1 - t0, t2 - t1, t3 - t2, n * R / (t2 - t1)))
    print('   AD gyro x at tau=1s: %.4g deg/hr-ish units' % (list(ad.values())[0][9, 0] if isinstance(ad, dict) else ad[9, 0]))
