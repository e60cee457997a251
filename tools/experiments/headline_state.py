"""Why is the headline launch (C2: 65 536 runs x 1000 samples, fp64, everything kept) 1.17 ms in some processes and 1.39 ms in
others on ONE box?  In one process: time the launch, free every buffer (hipFree, pool drained), allocate again under several
conditions (plain; behind a dummy allocation of D GB; accel+gyro+trajectory carved out of ONE allocation) and time again.
Prints one JSON line per condition; `clocks` = what rocm-smi reports right after the timed launches."""
import json
import os
import subprocess
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


def clocks():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=20).stdout
        keep = {}
        for l in out.splitlines():
            for k in ('fclk', 'mclk', 'sclk', 'socclk'):
                if k + ' clock' in l and '(' in l:
                    keep[k] = l.split('(')[-1].split(')')[0]
        return keep
    except Exception as e:      # noqa: BLE001
        return {'error': repr(e)[:80]}


def measure(ctx, job, reps=60):
    for _ in range(25):
        job.launch()
    ctx.sync()
    ms, mn = bench.time_launches(ctx, job.launch, reps, warm=0)
    return ms, mn


def main():
    ctx = ginsim.Context(0)
    fs, rf, R = 100.0, 1, int(os.environ.get('RUNS', '65536'))
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')

    def make():
        return ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=bench.SEED, keep_sensors=True, keep_traj=True)

    def report(tag, job, extra=None):
        ms, mn = measure(ctx, job)
        ptrs = {k: hex(v.ptr) for k, v in job._bufs.items() if hasattr(v, 'ptr') and k in ('imu', 'accel', 'traj_free', 'end_free')}
        d = {'condition': tag, 'kernel_ms': ms, 'kernel_ms_min': mn, 'frac': job.bytes_written() / (ms * 1e-3) / 8e12, 'ptrs': ptrs, 'clocks': clocks()}
        d.update(extra or {})
        print(json.dumps(d), flush=True)

    conds = sys.argv[1:] or ['plain', 'plain', 'dummy4', 'dummy16', 'dummy64', 'plain']
    dummies = []
    for c in conds:
        if c.startswith('dummy'):
            gb = int(c[5:])
            dummies.append(ctx.malloc(gb << 30))     # stays allocated: shifts where the next buffers land
        elif c == 'undummy':
            for d in dummies:
                d.free()
            dummies = []
        elif c.startswith('sleep'):
            time.sleep(float(c[5:]))
        job = make()
        report(c, job)
        job.release()
        del job
        ctx.release_pool()


if __name__ == '__main__':
    main()
