"""The headline launch with its planes placed in chosen 96 GB thirds of the device memory (a 230 GB arena that starts near
physical 0: tools/exp_r05s.sh found the fast windows where the 7.9 GB block straddles arena offsets 96 and 192 GB)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'gnss-ins-sim_amd'))
sys.path.insert(0, ROOT)
os.environ['GINSIM_MALLOC_FLAGS'] = '4'
import ginsim          # noqa: E402
from ginsim import workloads   # noqa: E402
import bench           # noqa: E402

G = 1 << 30


def main():
    ctx = ginsim.Context(0)
    fs, rf, R = 100.0, 1, 65536
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    arena = ctx.malloc(230 * G)
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=bench.SEED, keep_sensors=True, keep_traj=True)
    plane = job.n * R * 8
    p = job.params
    cases = [('all in third 0', 10, 10 + 3 * plane / G, 10 + 6 * plane / G),
             ('sensors third 0, trajectory third 1', 10, 10 + 3 * plane / G, 100),
             ('sensors third 0, trajectory third 2', 10, 10 + 3 * plane / G, 200),
             ('accel third 0, gyro third 1, trajectory third 2', 10, 100, 200),
             ('accel third 0, gyro third 1, trajectory across 1|2 (4.4 + 4.6 planes)', 10, 100, 192 - 4.4 * plane / G),
             ('accel third 0, gyro third 2, trajectory across 0|1', 10, 200, 96 - 4.5 * plane / G),
             ('sensors across 0|1, trajectory across 1|2', 96 - 3 * plane / G, 96, 192 - 4.5 * plane / G),
             ('all across 0|1 (7.5 planes each side)', 96 - 7.5 * plane / G, 96 - 4.5 * plane / G, 96 - 1.5 * plane / G),
             ('all across 0|1 (3 | 12)', 96 - 3 * plane / G, 96, 96 + 3 * plane / G),
             ('all in third 1', 110, 110 + 3 * plane / G, 110 + 6 * plane / G)]
    for tag, oa, og, ot in cases:
        al = lambda x: arena.ptr + (int(x * G) // 4096) * 4096
        p.out_accel, p.out_gyro, p.out_traj[0] = al(oa), al(og), al(ot)
        for _ in range(15):
            job.launch()
        ctx.sync()
        ms, mn = bench.time_launches(ctx, job.launch, 40, warm=0)
        print(json.dumps({'placement': tag, 'kernel_ms': round(ms, 4), 'kernel_ms_min': round(mn, 4), 'frac': round(job.bytes_written() / (ms * 1e-3) / 8e12, 3)}), flush=True)


if __name__ == '__main__':
    main()
