#!/usr/bin/env python3
"""Time the UNMODIFIED reference's own CPU path on BASELINE config 1 (90-degree turn @100 Hz, 'mid-accuracy' 6-axis IMU,
ref_frame 1, FreeIntegration) and write the host-stamped record profiles/reference_cpu.json:

  * one core (the reference is single-threaded): Sim.run(R) end to end, split into noise generation (pathgen.acc_gen +
    gyro_gen), the plugin (FreeIntegration.run) and the rest of run(), plus Sim.results(err_stats_start=-1) -- the split of
    BASELINE.md section 2, taken with timers wrapped around the reference's own functions;
  * all host cores: multiprocessing with P = os.cpu_count() workers, each running the
    reference's Sim.run(R / P) with a seed of its own (ins_sim.py:164-192, 490-506 is the loop being spread), aggregate
    sample*MC/s from the common start to the last worker's end.

Runs only where the reference checkout exists (the build container); bench.py calls it there (--json) and quotes the committed
record on the GPU box, where the reference cannot be imported.

    python tools/time_reference.py [--runs R] [--procs P] [--json] [--no-write]
"""
import argparse
import datetime
import json
import multiprocessing
import os
import platform
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('GNSS_INS_SIM_REFERENCE', '/root/reference')
N_SAMPLES = 1000


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def _setup():
    """Import the reference (never this repository's drop-in of the same package name) without writing bytecode into it."""
    sys.dont_write_bytecode = True
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path[:] = [REF] + [p for p in sys.path if 'gnss-ins-sim_amd' not in p]
    import numpy as np
    from gnss_ins_sim.sim import imu_model, ins_sim
    from gnss_ins_sim.pathgen import pathgen
    from demo_algorithms import free_integration
    assert os.path.abspath(ins_sim.__file__).startswith(os.path.abspath(REF)), ins_sim.__file__
    return np, imu_model, ins_sim, pathgen, free_integration


def _make_sim(mods, seed):
    np, imu_model, ins_sim, _, free_integration = mods
    csv = REF + '/demo_motion_def_files/motion_def-90deg_turn.csv'
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    np.random.seed(seed)
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    return ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=free_integration.FreeIntegration(ini))


def _quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def one_core(runs):
    mods = _setup()
    _, _, _, pathgen, free_integration = mods
    acc = {'noise': 0.0, 'plugin': 0.0}

    def timed(fn, key):
        def wrapper(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[key] += time.perf_counter() - t
        return wrapper
    pathgen.acc_gen = timed(pathgen.acc_gen, 'noise')
    pathgen.gyro_gen = timed(pathgen.gyro_gen, 'noise')
    free_integration.FreeIntegration.run = timed(free_integration.FreeIntegration.run, 'plugin')
    sim = _make_sim(mods, 2024)
    t0 = time.perf_counter()
    _quiet(sim.run, runs)
    t_run = time.perf_counter() - t0
    t0 = time.perf_counter()
    _quiet(sim.results, err_stats_start=-1)
    t_res = time.perf_counter() - t0
    units = runs * N_SAMPLES
    return {'value': units / t_run, 'unit': 'sample*MC/s', 'cores': 1, 'runs': runs, 'run_wall_s': t_run,
            'split_s': {'noise_generation (pathgen.acc_gen + gyro_gen)': acc['noise'], 'plugin (FreeIntegration.run)': acc['plugin'],
                        'rest of Sim.run (att_quat, bookkeeping, path_gen)': t_run - acc['noise'] - acc['plugin'],
                        'Sim.results(err_stats_start=-1)': t_res},
            'noise_plus_plugin_sample_MC_per_s': units / (acc['noise'] + acc['plugin']),
            'sample': 'unmodified reference Sim.run(%d) on config 1 (90-degree turn @100 Hz, mid-accuracy, ref_frame 1), %.1f s, '
                      'single-threaded by construction' % (runs, t_run)}


def _worker(args):
    rank, runs, start_at = args
    mods = _setup()
    sim = _make_sim(mods, 2024 + rank)
    while time.time() < start_at:           # common start: the imports of all workers are over
        time.sleep(0.001)
    t0 = time.time()
    _quiet(sim.run, runs)
    return t0, time.time()


def all_cores(runs_total, procs):
    per = max(1, runs_total // procs)
    start_at = time.time() + 8.0            # workers import numpy / the reference first (a few seconds on a cold page cache)
    ctx = multiprocessing.get_context('spawn')
    with ctx.Pool(procs) as pool:
        spans = pool.map(_worker, [(r, per, start_at) for r in range(procs)], chunksize=1)
    t0, t1 = min(s[0] for s in spans), max(s[1] for s in spans)
    assert t0 >= start_at - 0.05, 'a worker was not ready at the common start: raise the lead time'
    return {'value': per * procs * N_SAMPLES / (t1 - t0), 'unit': 'sample*MC/s', 'cores': procs, 'runs': per * procs,
            'wall_s': t1 - t0,
            'sample': 'unmodified reference, multiprocessing with %d workers x Sim.run(%d) on config 1, distinct seeds, common '
                      'start to last end %.1f s' % (procs, per, t1 - t0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--runs', type=int, default=1000, help='runs of the one-core measurement (all cores: the same per worker / 4, at least 50)')
    ap.add_argument('--procs', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--json', action='store_true', help='print the record as one JSON line (bench.py reads it)')
    ap.add_argument('--no-write', action='store_true')
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, 'gnss_ins_sim')):
        sys.exit('the reference is not importable here (%s)' % REF)
    one = one_core(a.runs)
    per_worker = max(50, a.runs // 4) if a.runs >= 200 else max(2, a.runs // 4)
    many = all_cores(per_worker * a.procs, a.procs) if a.procs > 1 else None
    try:
        head = subprocess.run(['git', '-C', REF, 'rev-parse', 'HEAD'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              universal_newlines=True).stdout.strip()
    except OSError:
        head = ''
    out = dict(one)
    out.update({'all_cores': many, 'host': platform.node(), 'cpu': cpu_model(), 'logical_cpus': os.cpu_count(),
                'date': datetime.datetime.now(datetime.timezone.utc).strftime('%Y-%m-%dT%H:%MZ'),
                'python': platform.python_version(), 'numpy': __import__('numpy').__version__, 'reference_head': head,
                'made_by': 'tools/time_reference.py'})
    if not a.no_write:
        with open(os.path.join(REPO, 'profiles', 'reference_cpu.json'), 'w') as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write('\n')
    print(json.dumps(out) if a.json else json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
