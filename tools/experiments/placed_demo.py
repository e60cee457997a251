"""Placed arena on the headline job (round 6): build the arena, print what the search found, time C2's launch with placed and with
plain hipMalloc planes, interleaved; then grow the arena and time C4's share."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, 'gnss-ins-sim_amd'), ROOT]
import ginsim                       # noqa: E402
from ginsim import workloads        # noqa: E402
import bench                        # noqa: E402


def main():
    ctx = ginsim.Context(0)
    fs, rf = 100.0, 1
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    for R, prec in ((65536, 'f64'), (131072, 'f64'), (65536, 'f32')):
        jobs = {}
        for tag, placed in (('asis', False), ('placed', True)):
            t0 = time.perf_counter()
            jobs[tag] = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=bench.SEED, keep_sensors=True,
                                             keep_traj=True, placed=placed, precision=prec)
            print(json.dumps({'runs': R, 'precision': prec, 'job': tag, 'construct_s': round(time.perf_counter() - t0, 3), 'note': ctx.placed_note}), flush=True)
        info = ctx.placed_info()
        print(json.dumps({k: info[k] for k in ('available', 'classes', 'searches', 'mapped_bytes', 'used_bytes', 'stripes_of_class', 'chunks_created',
                                               'chunks_ambiguous', 'probes', 'peak_held_bytes', 'search_seconds', 'last_search_seconds', 'anchor_ms',
                                               'stripe_classes')}), flush=True)
        for rep in range(3):
            for tag in ('asis', 'placed'):
                ms, mn = bench.time_launches(ctx, jobs[tag].launch, 30)
                print(json.dumps({'runs': R, 'precision': prec, 'job': tag, 'kernel_ms': round(ms, 4), 'min': round(mn, 4),
                                  'frac': round(jobs[tag].bytes_written() / (ms * 1e-3) / 8e12, 3)}), flush=True)
        same = bool((jobs['asis'].end_errors('free') == jobs['placed'].end_errors('free')).all())
        print(json.dumps({'same_end_errors': same}), flush=True)
        for j in jobs.values():
            j.release()
    ctx.close()


if __name__ == '__main__':
    main()
