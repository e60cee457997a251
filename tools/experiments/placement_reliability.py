"""Round 6, VERDICT r05 item 1: the headline launch (C2: 65 536 runs x 1000 samples, fp64, 15 planes materialised) with PLACED planes
(the library default, what an unconfigured Sim gets) and with plain hipMalloc planes, in FRESH processes.  Each process: build both
jobs, pre-warm by time, 200 launches of each back to back, HIP events around every launch.

    python tools/experiments/placement_reliability.py --processes 6        # prints one JSON line per process + the box's identity
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def child():
    import time
    sys.path[:0] = [os.path.join(ROOT, 'gnss-ins-sim_amd'), ROOT]
    import ginsim
    from ginsim import workloads
    import bench
    ctx = ginsim.Context(0)
    fs, rf, R = 100.0, 1, 65536
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    out = {}
    hold = int(os.environ.get('HOLD_GB', '0'))        # what else the process holds first (moves where hipMalloc puts the plain planes)
    spacer = ctx.malloc(hold << 30) if hold else None
    for tag, placed in (('placed', True), ('asis', False)):
        t0 = time.perf_counter()
        job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=bench.SEED, keep_sensors=True,
                                   keep_traj=True, placed=placed)
        built = time.perf_counter() - t0
        ms, mn = bench.time_launches(ctx, job.launch, 200)
        out[tag] = {'kernel_ms_avg': round(ms, 4), 'kernel_ms_min': round(mn, 4), 'frac': round(job.bytes_written() / (ms * 1e-3) / 8e12, 4),
                    'construct_s': round(built, 3), 'placed_regions': job.placement()['placed']}
        if placed:
            a = job.placement()['arena']
            out['arena'] = {k: a[k] for k in ('classes', 'stripes_of_class', 'chunks_created', 'chunks_ambiguous', 'probes', 'peak_held_bytes',
                                              'search_seconds', 'anchor_ms', 'stripe_classes')}
            out['note'] = ctx.placed_note
        job.release()
    out['hold_gb'] = hold
    print(json.dumps(out), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--processes', type=int, default=6)
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child()
    ident = {}
    try:
        ident['hostname'] = os.uname().nodename
        out = subprocess.run(['rocm-smi', '--showuniqueid', '--showbus', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=30).stdout
        ident['rocm_smi'] = json.loads(out)
    except Exception as e:                  # noqa: BLE001
        ident['rocm_smi_error'] = repr(e)[:100]
    print(json.dumps({'box': ident}), flush=True)
    holds = [0, 0, 0, 24, 60, 120]
    for k in range(args.processes):
        env = dict(os.environ, HOLD_GB=str(holds[k % len(holds)]))
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, check=False)


if __name__ == '__main__':
    main()
