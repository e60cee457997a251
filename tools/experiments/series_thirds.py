"""Few-run generation (32 x 1 440 000) with its two output halves (accel series, gyro series: 1.1 GB each) placed in the same or in
different 96 GB thirds of the device memory (230 GB contiguous arena, boundary at arena offset 96 GB: tools/exp_r05s.sh); and the
Allan call reading them."""
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
    half = 3 * n * runs * 8
    p = job.params
    S = 6 * runs
    for tag, oa, og in (('both halves in third 0', 10 * G, 10 * G + half), ('accel in third 0, gyro in third 1', 96 * G - half, 96 * G),
                        ('both halves in third 1', 110 * G, 110 * G + half), ('accel in third 0, gyro in third 2', 10 * G, 200 * G)):
        p.out_accel, p.out_gyro = arena.ptr + oa, arena.ptr + og
        for _ in range(6):
            job.launch()
        ctx.sync()
        ms, mn = bench.time_launches(ctx, job.launch, 20)
        row = {'placement': tag, 'generation_ms': round(ms, 4), 'generation_ms_min': round(mn, 4)}
        if og == oa + half:         # contiguous: the Allan call can read it
            x = ginsim.engine.DeviceView(arena, oa, 2 * half, 'series')
            for _ in range(30):
                ginsim.allan_var(ctx, x, n, S, n, fs)
            t = []
            for _ in range(20):
                ctx.timer_begin()
                ginsim.allan_var(ctx, x, n, S, n, fs)
                t.append(ctx.timer_end())
            row['allan_call_ms'] = round(sum(t) / len(t), 4)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
