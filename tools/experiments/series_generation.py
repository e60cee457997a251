"""Time the few-run sensor generation (BASELINE config 5's first half: 32 runs x 1 440 000 samples, accel + gyro, series-major)
with the library named by $GINSIM_LIB (A/B builds).  Prints one JSON line."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'gnss-ins-sim_amd'))
sys.path.insert(0, ROOT)
import numpy as np     # noqa: E402
import ginsim          # noqa: E402
from ginsim import workloads   # noqa: E402
import bench           # noqa: E402


def main(runs=32, seconds=3600.0, fs=400.0, reps=20):
    ctx = ginsim.Context(0)
    text = open(workloads.profile_path('static_1800s')).read().split('\n')
    ini, _ = workloads.parse_motion('\n'.join(text[:4]))
    seg = np.array([[1.0, 0, 0, 0, 0, 0, 0, seconds, 0.0]])
    raw = ginsim.pathgen(ini, seg, fs, 0.0, workloads.HIGH_MOBILITY, 1)
    truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
             'ref_pos': raw['nav'][:, 1:4], 'ref_vel': raw['nav'][:, 4:7], 'ref_att': raw['nav'][:, 7:10]}
    n = truth['ref_accel'].shape[0]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ctx, fs, 1, truth, acc, gyr, None, runs=runs, algos=(), seed=bench.SEED, keep_sensors=True)
    job.run()
    for _ in range(5):
        job.launch()
    ctx.sync()
    ms, mn = bench.time_launches(ctx, job.launch, reps)
    ids = np.array([0, runs - 1])
    a = job.sensors('accel', ids)
    g = job.sensors('gyro', ids)
    print(json.dumps({'lib': os.path.basename(ginsim._lib.LIB_PATH), 'kernel': job.kernel_name(), 'runs': runs, 'n': int(n),
                      'generation_ms': ms, 'generation_ms_min': mn, 'hbm_frac': 48.0 * runs * n / (ms * 1e-3) / 8e12,
                      'checksum_accel': float(np.abs(a).sum()), 'checksum_gyro': float(np.abs(g).sum())}))


if __name__ == '__main__':
    main()
