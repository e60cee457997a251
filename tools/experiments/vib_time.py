import sys, time, gc
sys.path.insert(0, 'gnss-ins-sim_amd'); sys.path.insert(0, 'examples')
import numpy as np
import demo_vibration as d
from gnss_ins_sim.sim import imu_model, ins_sim
from demo_algorithms import free_integration
import ginsim
ini = np.genfromtxt(d.MOTION, delimiter=',', skip_header=1, max_rows=1); ini[0:2] *= d.D2R; ini[6:9] *= d.D2R
for rep in range(2):
    for label, env in d.ENVS:
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
        t0 = time.perf_counter()
        sim = ins_sim.Sim([100.0, 0.0, 0.0], d.MOTION, ref_frame=1, imu=imu, mode=None, env=env, algorithm=free_integration.FreeIntegration(ini), seed=2024)
        t1 = time.perf_counter()
        sim.run(65536)
        t2 = time.perf_counter()
        info = ginsim.default_context().placed_info()
        print('%d %-30s ctor %.1f ms run %.1f ms  searches %d mapped %.1f GiB used %.1f GiB search_s %.2f' % (rep, label[:30], (t1-t0)*1e3, (t2-t1)*1e3, info['searches'], info['mapped_bytes']/2**30, info['used_bytes']/2**30, info['search_seconds']), flush=True)
