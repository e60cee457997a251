"""The Allan call (192 series x 1 440 000 samples, 2.2 GB read once) on a copy of the same series at offsets 0, 4, 8, ... GB of a
230 GB physically contiguous arena: does the call's time follow the position of the series in the device memory?"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'gnss-ins-sim_amd'))
sys.path.insert(0, ROOT)
os.environ['GINSIM_MALLOC_FLAGS'] = '4'
import numpy as np     # noqa: E402
import ginsim          # noqa: E402
from ginsim import workloads   # noqa: E402
from ginsim._lib import check  # noqa: E402
import bench           # noqa: E402

G = 1 << 30


def main(runs=32, seconds=3600.0, fs=400.0):
    ctx = ginsim.Context(0)
    text = open(workloads.profile_path('static_1800s')).read().split('\n')
    ini, _ = workloads.parse_motion('\n'.join(text[:4]))
    seg = np.array([[1.0, 0, 0, 0, 0, 0, 0, seconds, 0.0]])
    raw = ginsim.pathgen(ini, seg, fs, 0.0, workloads.HIGH_MOBILITY, 1)
    truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
             'ref_pos': raw['nav'][:, 1:4], 'ref_vel': raw['nav'][:, 4:7], 'ref_att': raw['nav'][:, 7:10]}
    n = truth['ref_accel'].shape[0]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    arena = ctx.malloc(230 * G)
    job = ginsim.MonteCarloJob(ctx, fs, 1, truth, acc, gyr, None, runs=runs, algos=(), seed=bench.SEED, keep_sensors=True)
    job.run()
    S = 6 * runs
    size = 8 * S * n
    src = job.buffer('accel')
    for _ in range(40):
        ginsim.allan_var(ctx, src, n, S, n, fs)
    out = []
    step = float(os.environ.get('STEP_GB', '4'))
    off = 0.0
    while int(off * G) + size <= 230 * G:
        x = ginsim.engine.DeviceView(arena, (int(off * G) // 4096) * 4096, size, 'series')
        check(ginsim.lib.ginsim_runs_to_series(ctx.handle, src.ptr, 1, S * n, 1, x.ptr))        # C = 1, R = 1: a plain device copy
        for _ in range(6):
            ginsim.allan_var(ctx, x, n, S, n, fs)
        t = []
        for _ in range(12):
            ctx.timer_begin()
            ginsim.allan_var(ctx, x, n, S, n, fs)
            t.append(ctx.timer_end())
        out.append((off, round(sum(t) / len(t), 4)))
        off += step
    print(json.dumps({'arena': hex(arena.ptr), 'offset_gb__allan_call_ms': out}))


if __name__ == '__main__':
    main()
