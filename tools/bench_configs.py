#!/usr/bin/env python3
"""Measure every BASELINE.json configuration on one MI355X and write profiles/<tag>_configs.json.
  C2  90-degree turn @100 Hz, 65 536 runs, fp64 (materialised and stats-only)
  C3  long_drive @200 Hz (n = 193 036), ref_frame 0, 262 144 runs, stats-only (trajectories would be 6 TB)
  C4  per-GPU share of 1 048 576 runs over 8 GPUs = 131 072 runs of the C2 profile
  C5  fp32 kernel on the C2 profile + on-device Allan variance of 3600 s @ 400 Hz series
"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ctx = ginsim.Context(0)
acc, gyr = workloads.imu_grade('mid-accuracy')
out = {'device': ctx.name(), 'hbm_peak_GBps': 8000.0, 'rows': []}


def measure(name, profile, fs, rf, R, keep, precision='f64', reps=30, gps=False):
    ini, truth, raw = workloads.truth_from_profile(profile, fs, rf, fs_gps=10.0 if gps else 0.0, gps=gps)
    n = truth['ref_accel'].shape[0]
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=keep, keep_traj=keep,
                               precision=precision)
    job.run()
    ts = []
    for _ in range(reps):
        ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
    st = job.stats('free')
    ms = float(np.median(ts))
    unit = (120 if precision == 'f64' else 60) if keep else 0
    alg = unit * R * n + 72 * R
    row = {'config': name, 'profile': profile, 'fs': fs, 'ref_frame': rf, 'runs': R, 'samples_per_run': n, 'precision': precision,
           'materialised': keep, 'kernel_ms_median': ms, 'kernel_ms_min': float(min(ts)), 'sample_MC_per_s': R * n / ms * 1e3,
           'algorithmic_bytes': alg, 'achieved_GBps': alg / ms / 1e6, 'frac_of_hbm_peak': alg / ms / 1e6 / 8000.0,
           'att_std_deg': (st.std[:3] * 180 / np.pi).tolist(), 'vel_std': st.std[6:9].tolist()}
    print(json.dumps(row), flush=True)
    out['rows'].append(row)
    job.release()


measure('C2 fp64 materialised', 'turn_90deg', 100.0, 1, 65536, True)
measure('C2 fp64 stats-only', 'turn_90deg', 100.0, 1, 65536, False)
measure('C4 per-GPU share (131072 runs) fp64 materialised', 'turn_90deg', 100.0, 1, 131072, True)
measure('C4 per-GPU share (131072 runs) fp64 stats-only', 'turn_90deg', 100.0, 1, 131072, False)
measure('C2 profile, 262144 runs fp64 materialised', 'turn_90deg', 100.0, 1, 262144, True)
measure('C5 fp32 materialised', 'turn_90deg', 100.0, 1, 65536, True, precision='f32')
measure('C5 fp32 stats-only', 'turn_90deg', 100.0, 1, 65536, False, precision='f32')
measure('C5 fp32 262144 runs materialised', 'turn_90deg', 100.0, 1, 262144, True, precision='f32')
measure('C3 long_drive 262144 runs stats-only', 'long_drive', 200.0, 0, 262144, False, reps=2, gps=True)
measure('C3 long_drive 262144 runs stats-only fp32', 'long_drive', 200.0, 0, 262144, False, precision='f32', reps=2, gps=True)


def measure_given(name, rf, R):
    """FreeIntegration.run alone for a whole batch: given-sensors kernel on device-resident series (48 B read + 72 B written)."""
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    n = truth['ref_accel'].shape[0]
    gen = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=True).run()
    rep = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, None, None, ini, runs=R, keep_traj=True,
                               given={'gyro': gen.buffer('gyro'), 'accel': gen.buffer('accel')}).run()
    ts = []
    for _ in range(30):
        ctx.timer_begin(); rep.launch(); ts.append(ctx.timer_end())
    ms = float(np.median(ts))
    alg = 120 * R * n + 72 * R
    row = {'config': name, 'profile': 'turn_90deg', 'fs': 100.0, 'ref_frame': rf, 'runs': R, 'samples_per_run': n, 'precision': 'f64',
           'kernel': rep.kernel_name(), 'kernel_ms_median': ms, 'kernel_ms_min': float(min(ts)), 'sample_MC_per_s': R * n / ms * 1e3,
           'algorithmic_bytes': alg, 'achieved_GBps': alg / ms / 1e6, 'frac_of_hbm_peak': alg / ms / 1e6 / 8000.0}
    print(json.dumps(row), flush=True)
    out['rows'].append(row)
    rep.release(); gen.release()


measure_given('mechanisation only (given sensors on device), C2 shape', 1, 65536)
measure_given('mechanisation only (given sensors on device), 131072 runs', 1, 131072)
measure_given('mechanisation only (given sensors on device), ref_frame 0', 0, 65536)
# Allan (C5)
S, n, fs = 192, 1440000, 400.0
x = np.random.default_rng(0).normal(size=(S, n))
buf = ctx.upload(x)
ginsim.allan_var(ctx, buf, n, S, n, fs)
ts = []
for _ in range(8):
    ctx.timer_begin(); avar, tau = ginsim.allan_var(ctx, buf, n, S, n, fs); ts.append(ctx.timer_end())
ms = float(np.median(ts))
row = {'config': 'C5 Allan variance, %d series x %d samples (3600 s @ 400 Hz), 46 tau' % (S, n), 'ms_median': ms,
       'algorithmic_bytes': 8.0 * S * n, 'achieved_GBps': 8.0 * S * n / ms / 1e6, 'frac_of_hbm_peak': 8.0 * S * n / ms / 1e6 / 8000.0,
       'samples_per_s': S * n / ms * 1e3}
print(json.dumps(row), flush=True)
out['rows'].append(row)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(REPO, 'gpurun_out', tag + '_configs.json'), 'w'), indent=1)
