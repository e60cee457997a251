"""The headline launch with its 15 output planes back to back (7.9 GB) at offsets 0, 2, 4, ... GB of ONE physically contiguous
arena: is the launch time a property of WHERE in the device memory the planes lie?  One JSON line."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'gnss-ins-sim_amd'))
sys.path.insert(0, ROOT)
os.environ['GINSIM_MALLOC_FLAGS'] = os.environ.get('ARENA_FLAGS', '4')
import ginsim          # noqa: E402
from ginsim import workloads   # noqa: E402
import bench           # noqa: E402


def main():
    ctx = ginsim.Context(0)
    fs, rf, R = 100.0, 1, 65536
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    pre = [ctx.malloc(int(g) << 30) for g in os.environ.get('PRE_GB', '').split(',') if g]
    arena_gb = int(os.environ.get('ARENA_GB', '96'))
    step_gb = float(os.environ.get('STEP_GB', '2'))
    arena = ctx.malloc(arena_gb << 30)
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=bench.SEED, keep_sensors=True, keep_traj=True)
    plane = job.n * R * 8
    p = job.params
    out, off = [], 0.0
    while int(off * (1 << 30)) + 15 * plane <= (arena_gb << 30):
        b = arena.ptr + int(off * (1 << 30))
        p.out_accel, p.out_gyro, p.out_traj[0] = b, b + 3 * plane, b + 6 * plane
        for _ in range(12):
            job.launch()
        ctx.sync()
        ms, mn = bench.time_launches(ctx, job.launch, 24, warm=0)
        out.append((off, round(ms, 4)))
        off += step_gb
    print(json.dumps({'arena': hex(arena.ptr), 'arena_gb': arena_gb, 'pre_gb': os.environ.get('PRE_GB', ''), 'offset_gb__kernel_ms': out}))


if __name__ == '__main__':
    main()
