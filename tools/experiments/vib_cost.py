#!/usr/bin/env python3
"""What the vibration term of Sim(env=...) costs: BASELINE config 2's launch (65 536 runs x 1000 samples, ref_frame 1, free
integration, trajectories kept) without vibration (wave-specialised simple-model kernel), and with a random / sinusoidal model
on both sensors (vibration variant of the plain general-model kernel).  One JSON line per case."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads

ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
acc, gyr = workloads.imu_grade('mid-accuracy')
R = int(os.environ.get('RUNS', 65536))
cases = {
    'none': (None, None),
    'random': ({'type': 'random', 'x': 0.3, 'y': 0.1, 'z': 0.2}, {'type': 'random', 'x': 2e-3, 'y': 1e-3, 'z': 3e-3}),
    'sinusoidal': ({'type': 'sinusoidal', 'x': 0.3, 'y': 0.1, 'z': 0.2, 'freq': 7.0}, {'type': 'sinusoidal', 'x': 5e-3, 'y': 2e-3, 'z': 1e-3, 'freq': 0.9}),
}
for name, (va, vg) in cases.items():
    for keep in (True, False):
        job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=7, keep_sensors=keep, keep_traj=keep, vib_accel=va, vib_gyro=vg)
        for _ in range(10):
            job.launch()
        ts = []
        for _ in range(20):
            ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
        st = job.stats('free')
        print(json.dumps({'vibration': name, 'kept': keep, 'kernel': job.kernel_name(), 'ms_min': min(ts), 'ms_avg': sum(ts) / len(ts),
                          'sample_MC_per_s': R * 1000 / (min(ts) * 1e-3), 'att_std_deg': (st.std[:3] * 180 / np.pi).tolist()}))
        job.release()
