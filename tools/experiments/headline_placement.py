"""The headline launch with its 15 output planes placed by hand inside ONE physically contiguous arena: the sensor planes
(6 x 500 MiB) at the arena's start, the trajectory planes (9 x 500 MiB) `gap` bytes behind them.  One JSON line per gap."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'gnss-ins-sim_amd'))
sys.path.insert(0, ROOT)
os.environ['GINSIM_MALLOC_FLAGS'] = os.environ.get('ARENA_FLAGS', '4')
import numpy as np     # noqa: E402
import ginsim          # noqa: E402
from ginsim import workloads   # noqa: E402
import bench           # noqa: E402


def main():
    ctx = ginsim.Context(0)
    fs, rf, R = 100.0, 1, 65536
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    pre = [ctx.malloc(int(g) << 30) for g in os.environ.get('PRE_GB', '').split(',') if g]      # shifts the arena's base
    arena = ctx.malloc(40 << 30)
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=bench.SEED, keep_sensors=True, keep_traj=True)
    n = job.n
    plane = n * R * 8
    p = job.params
    own = (p.out_accel, p.out_gyro, p.out_traj[0])

    def timed(tag, extra):
        for _ in range(20):
            job.launch()
        ctx.sync()
        ms, mn = bench.time_launches(ctx, job.launch, 40, warm=0)
        print(json.dumps(dict({'placement': tag, 'kernel_ms': ms, 'kernel_ms_min': mn, 'frac': job.bytes_written() / (ms * 1e-3) / 8e12}, **extra)), flush=True)

    timed('own allocations', {'arena': hex(arena.ptr)})
    gaps = [0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 100 << 20, 128 << 20, 250 << 20, 256 << 20, 1 << 30, (1 << 30) + (12 << 20), 8 << 30, 16 << 30]
    for g in gaps:
        p.out_accel, p.out_gyro, p.out_traj[0] = arena.ptr, arena.ptr + 3 * plane, arena.ptr + 6 * plane + g
        timed('arena', {'gap': g})
    # and the whole group shifted inside the arena (same relative placement, another absolute one)
    for sh in (1 << 20, 64 << 20, 1 << 30, 5 << 30, 17 << 30):
        p.out_accel, p.out_gyro, p.out_traj[0] = arena.ptr + sh, arena.ptr + sh + 3 * plane, arena.ptr + sh + 6 * plane
        timed('arena shifted', {'shift': sh})
    p.out_accel, p.out_gyro, p.out_traj[0] = own
    timed('own allocations', {})


if __name__ == '__main__':
    main()
