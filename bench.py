#!/usr/bin/env python3
"""Benchmark of the Monte-Carlo strapdown-INS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of Monte-Carlo runs on every GPU:
fused kernel (noise injection + free-integration mechanisation + end-point error, sensors and
trajectories materialised in HBM exactly as the reference's Sim holds them after run()) -> on-device
end-point statistics -> (N > 1) one all-reduce of the per-GPU statistics records -> merged mean/std/max.
The host-side part of a step (waiting for the 28-double record, the all-reduce, the Chan merge) is done while the
next batches integrate (the record one step later, the non-blocking all-reduce two steps later): K timed steps =
K launches + K reductions + K exchanges, all inside the timed region.

Workload (config.workload names it):
  N = 1   BASELINE.json configs[1] (C2): 90-degree-turn profile @100 Hz (n = 1000), 'mid-accuracy' 6-axis IMU,
          ref_frame = 1, 65 536 runs, fp64 -- the configuration the metric is quoted on.
  N > 1   BASELINE.json configs[3] (C4): the same profile, 131 072 runs per GPU (1 048 576 over 8 GPUs), one RCCL
          all-reduce of the statistics records per step (backend nccl).  Weak scaling: every rank integrates its own
          runs (global run ids are disjoint, the Philox counter carries the global id).

`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) re-executes itself under torch.distributed.run with
N ranks on 127.0.0.1, so the driver's command line works for every N.  Before the counted --warmup steps the headline
kernel is pre-warmed by TIME (launches of this kernel get faster for the first 20-30 ms of back-to-back execution:
clock / power management settling), reported as `prewarm_ms`.  At N > 1 rank 0 first measures `per_gpu_single`: the SAME
per-GPU load (131 072 runs) on one GPU with the other ranks idle, so that scaling efficiency compares like with like.

The JSON line (rank 0) also carries, measured in the same process after the timed region (N = 1 only):
  roofline             dominant kernel: algorithmic bytes / HIP-event launch time, vs 8 TB/s; `traffic` = HBM bytes per
                       launch from rocprofv3 PMC passes (WRITE_SIZE, FETCH_SIZE x 2 on gfx950) run live on this build,
                       or from profiles/pmc_traffic.json when that file carries this build's library hash, else null
  configs[]            the other BASELINE configurations as legs: C3 (long_drive @200 Hz, 262 144 runs, stats-only),
                       C4's per-GPU share, C5 fp32, Allan end-to-end (3600 s @ 400 Hz), the mechanisation alone, and C2 in a
                       vibration environment (random; PSD arrays)
  cpu_baseline         the C port of the same path on the host cores (bounded sample) + the reference's own Python
                       path (timed here when /root/reference is importable, otherwise the BASELINE.md figure, labelled)
Inputs (truth, parameters) are resident in HBM before every timed region.
"""
import argparse
import hashlib
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_SAMPLE_MC = 120   # accel3 + gyro3 + att3 + pos3 + vel3 doubles, all writes
SEED = 20260923


def lib_hash():
    import ginsim
    with open(ginsim.LIB_PATH, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def roofline(alg_bytes, ms, kernel, traffic=None, **extra):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    out = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
           'traffic': traffic, 'kernel': kernel, 'kernel_ms_avg': ms, 'algorithmic_bytes_per_launch': alg_bytes}
    out.update(extra)
    return out


WARM_MS = 40.0      # launches of one kernel get faster for the first 20-30 ms of back-to-back execution (clock / power
                    # management settling: 1.65 -> 1.38 ms for the fused kernel, 0.61 -> 0.56 ms for the Allan call); the
                    # headline has its --warmup steps, the legs warm up by TIME


def time_launches(ctx, launch, reps, warm=2):
    """(average, minimum) HIP-event time of `reps` launches issued BACK TO BACK on the context's stream, as the timed
    region of the headline issues them: an event pair around every launch, no host synchronisation in between.
    Untimed warm-up first: `warm` launches, and more until WARM_MS of kernel time have gone by (40 launches at most)."""
    spent, done = 0.0, 0
    while done < warm or (warm > 0 and spent < WARM_MS and done < 40):
        ctx.event_record(0)
        launch()
        ctx.event_record(1)
        spent += ctx.event_elapsed(0, 1)
        done += 1
    for i in range(reps):
        ctx.event_record(2 * i)
        launch()
        ctx.event_record(2 * i + 1)
    ms = [ctx.event_elapsed(2 * i, 2 * i + 1) for i in range(reps)]
    return sum(ms) / len(ms), min(ms)


# --------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(fs, rf, ini, truth, acc, gyr, budget_s):
    """oracle/c/ginsim_oracle.c (kind 'port') on the host cores, same workload, bounded sample."""
    import ctypes
    from oracle import c_oracle
    cores = os.cpu_count() or 1
    c_oracle.lib()
    try:
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(cores)
    except OSError:
        cores = 1
    n = truth['ref_accel'].shape[0]
    t0 = time.perf_counter()
    c_oracle.mc_run(SEED, 0, 64 * cores, fs, rf, truth, acc, gyr, ini)
    probe = time.perf_counter() - t0
    runs = int(max(64 * cores, min(2_000_000, budget_s / max(probe, 1e-6) * 64 * cores)))
    t0 = time.perf_counter()
    c_oracle.mc_run(SEED, 0, runs, fs, rf, truth, acc, gyr, ini)
    dt = time.perf_counter() - t0
    out = {'value': runs * n / dt, 'unit': 'sample*MC/s', 'cores': cores, 'kind': 'port',
           'sample': '%d runs x %d samples of the same workload through oracle/c/ginsim_oracle.c '
                     '(OpenMP over runs, %.1f s)' % (runs, n, dt)}
    out['reference_python'] = reference_python_baseline(min(budget_s, 15.0))
    return out


def reference_python_baseline(budget_s):
    """The reference's own CPU path on config 1: one core (it is single-threaded; split into noise generation / plugin / rest
    as BASELINE.md section 2 does) AND all host cores (multiprocessing over Sim.run(R / P)).  Timed here by
    tools/time_reference.py when the reference is importable (the build container), otherwise the committed, host-stamped
    record of that tool (profiles/reference_cpu.json), labelled as quoted."""
    # where the unmodified reference lies: $GNSS_INS_SIM_REFERENCE (the variable the drop-in's fall-through uses; the way to time
    # the reference BESIDE the GPU on a GPU host -- README "reference CPU baseline"), else the build container's /root/reference
    ref = os.environ.get('GNSS_INS_SIM_REFERENCE') or '/root/reference'
    if os.path.isfile(os.path.join(ref, 'gnss_ins_sim', 'sim', 'ins_sim.py')) and os.path.isdir(os.path.join(ref, 'demo_motion_def_files')):
        runs = max(20, int(budget_s * 4.5e4 / 1000))
        try:
            out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'time_reference.py'), '--runs', str(runs), '--json',
                                  '--no-write'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=40 * budget_s + 120,
                                 universal_newlines=True, cwd=tempfile.gettempdir(),
                                 env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1', GNSS_INS_SIM_REFERENCE=ref))
            rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
            rec['kind'] = 'reference'
            rec['timed_here'] = 'the unmodified reference at %s, in this run, on this host' % ref
            return rec
        except Exception as e:                                     # noqa: BLE001 -- a baseline must not fail the bench
            return {'value': None, 'kind': 'reference', 'error': repr(e)[:200]}
    try:
        with open(os.path.join(REPO, 'profiles', 'reference_cpu.json')) as f:
            rec = json.load(f)
        rec['kind'] = 'quoted'
        rec['quoted_from'] = ('profiles/reference_cpu.json (host %s, %s): /root/reference does not exist on this host, so the '
                              'reference is not timed here' % (rec.get('host', '?'), rec.get('date', '?')))
        return rec
    except (OSError, ValueError, KeyError):
        return {'value': 4.76e4, 'unit': 'sample*MC/s', 'cores': 1, 'kind': 'quoted',
                'sample': 'BASELINE.md section 2: unmodified reference Sim.run(1000) on config 1 in the survey container (Xeon 2.1 GHz, '
                          '1 core; the reference is single-threaded); /root/reference does not exist on this host, so it is not timed here'}


# --------------------------------------------------------------------------------------------- PMC passes
SIMDS, XCDS, SPEC_CLOCK_HZ = 1024, 8, 2.4e9       # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 8 XCDs, 2400 MHz
PMC_PASSES = (('WRITE_SIZE',), ('FETCH_SIZE',),
              ('SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY',
               'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE'))


def short_name(kernel_name):
    k = kernel_name[5:] if kernel_name.startswith('void ') else kernel_name
    return k[:k.index('(')] if '(' in k else k


def pmc_live(timeout_s=150):
    """Counters of every kernel of this script's --pmc-child workload (same build; the headline, the given-sensors and the
    fp32 launch, a C3-shaped launch cut to 8192 samples with and without the online statistics, config 5's sensor generation
    and three Allan calls) from one rocprofv3 --pmc pass per counter group, each in a run of its own with --kernel-trace only
    (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ has 8 slots, GRBM its own).
    Returns {kernel (as rocprofv3 names it, without 'void' and arguments): {counter: {'avg', 'sum', 'n'}, 'dur_ns': avg}} or None."""
    import sqlite3
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None
    out = {}
    work = tempfile.mkdtemp(prefix='ginsim_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for i, group in enumerate(PMC_PASSES):
            d = os.path.join(work, 'pass%d' % i)
            cmd = [exe, '--pmc'] + list(group) + ['--kernel-trace', '-d', d, '-o', 'pmc', '--', sys.executable,
                                                   os.path.abspath(__file__), '--pmc-child']
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith('.db')]
            if not dbs:
                return None
            con = sqlite3.connect(dbs[0])
            for name, counter, cnt, avg, tot in con.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) "
                                                            "from counters_collection group by kernel_name, counter_name"):
                out.setdefault(short_name(name), {})[counter] = {'avg': avg, 'sum': tot, 'n': cnt}
            if i == len(PMC_PASSES) - 1:
                for name, dur in con.execute("select name, avg(end - start) from kernels group by name"):
                    out.setdefault(short_name(name), {})['dur_ns'] = dur
            con.close()
        return out or None
    except Exception:                                              # noqa: BLE001 -- profiling is evidence, not the product
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def pmc_traffic(pmc, kernels):
    """HBM bytes per launch of `kernels`: WRITE_SIZE and FETCH_SIZE are in KiB; FETCH_SIZE x 2 on gfx950
    (MI355X_MICROARCH.md, HBM section: a wide coalesced read is tallied at half its bytes)."""
    out = {}
    for k in kernels:
        c = (pmc or {}).get(k, {})
        if 'WRITE_SIZE' in c and 'FETCH_SIZE' in c:
            w, f = c['WRITE_SIZE']['avg'] * 1024.0, 2.0 * c['FETCH_SIZE']['avg'] * 1024.0
            out[k] = {'hbm_bytes_per_launch': w + f, 'write_bytes': w, 'fetch_bytes_corrected': f, 'dispatches': int(c['WRITE_SIZE']['n'])}
    return out or None


def pmc_traffic_of_call(pmc, prefix, calls_kernel):
    """HBM bytes per CALL of a multi-kernel entry point (ginsim_allan): the counters of every kernel whose name starts with
    `prefix`, summed over all their dispatches, divided by the number of calls (= dispatches of `calls_kernel`)."""
    if not pmc or calls_kernel not in pmc or 'WRITE_SIZE' not in pmc[calls_kernel]:
        return None
    calls = pmc[calls_kernel]['WRITE_SIZE']['n']
    w = sum(c['WRITE_SIZE']['sum'] for k, c in pmc.items() if k.startswith(prefix) and 'WRITE_SIZE' in c) * 1024.0 / calls
    f = sum(c['FETCH_SIZE']['sum'] for k, c in pmc.items() if k.startswith(prefix) and 'FETCH_SIZE' in c) * 2048.0 / calls
    return {'hbm_bytes_per_call': w + f, 'write_bytes': w, 'fetch_bytes_corrected': f, 'calls': int(calls)}


def valu_roofline(pmc, kernel, kernel_ms, scale, note):
    """The roofline of a COMPUTE-bound launch ("report VALU utilisation"): SIMD cycles spent issuing VALU
    instructions per second against SIMDs x spec clock.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip (it equals
    SQ_INSTS_VALU within 1 % on these kernels: a wave64 fp64-class instruction occupies its SIMD for four cycles), GRBM_GUI_ACTIVE
    cycles summed over the 8 XCDs; both come from a launch of the same kernel cut to 8192 samples (`scale` = full / cut steps),
    `kernel_ms` is the HIP-event time of the FULL launch in this process."""
    c = (pmc or {}).get(kernel, {})
    if 'SQ_ACTIVE_INST_VALU' not in c or 'GRBM_GUI_ACTIVE' not in c:
        return None
    if kernel_ms is None:                                          # a kernel of a multi-launch call: its own time in the counter pass
        if not c.get('dur_ns'):
            return None
        kernel_ms = c['dur_ns'] * 1e-6
    busy = 4.0 * c['SQ_ACTIVE_INST_VALU']['avg']                   # SIMD cycles, cut launch
    cyc = c['GRBM_GUI_ACTIVE']['avg'] / XCDS                       # shader cycles the cut launch took
    ach = busy * scale / (kernel_ms * 1e-3)
    out = {'bound': 'valu', 'achieved': ach, 'peak': SIMDS * SPEC_CLOCK_HZ, 'unit': 'VALU-busy SIMD-cycles/s',
           'frac': ach / (SIMDS * SPEC_CLOCK_HZ), 'traffic': None, 'kernel': kernel, 'kernel_ms_avg': kernel_ms,
           'valu_busy_at_the_clock_it_ran': busy / (SIMDS * cyc),
           'effective_clock_ghz_profiled': (cyc / (c['dur_ns'] * 1e-9) / 1e9) if c.get('dur_ns') else None,
           'counters_per_cut_launch': {k: v['avg'] for k, v in c.items() if isinstance(v, dict)}, 'note': note}
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE', {}).get('avg'):
        out['lds_bank_conflict_frac'] = c['SQ_LDS_BANK_CONFLICT']['avg'] / c['SQ_LDS_IDX_ACTIVE']['avg']
    return out


def pmc_traffic_file(build):
    """profiles/pmc_traffic.json, only if it was measured on THIS build of libginsim.so."""
    try:
        with open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')) as f:
            d = json.load(f)
        return d.get('kernels') if d.get('libginsim_sha256') == build else None
    except (OSError, ValueError):
        return None


def brief_placement(p):
    """MonteCarloJob.placement() without the arena's stripe map (the headline's top-level `placement` carries that once)."""
    if p is None:
        return None
    a = p.get('arena', {})
    return {'placed': p['placed'], 'unplaced': p['unplaced'], 'bytes': p['bytes'],
            'arena_mapped_bytes': a.get('mapped_bytes'), 'arena_stripes_of_class': a.get('stripes_of_class')}


# --------------------------------------------------------------------------------------------- legs (N = 1)
def leg_mechanisation(ginsim, ctx, job, fs, rf, truth, ini, R, n, traffic, reps=10):
    """FreeIntegration.run alone for the whole batch: the given-sensors kernel reads the accel/gyro series the last step
    materialised (48 B) and writes att/pos/vel (72 B) per sample*MC -- the HBM-bound piece of the path."""
    rep = ginsim.MonteCarloJob(ctx, fs, rf, truth, None, None, ini, runs=R, keep_traj=True,
                               given={'gyro': job.buffer('gyro'), 'accel': job.buffer('accel')})
    rep.run()
    placed = rep.placement()
    avg, mn = time_launches(ctx, rep.launch, reps)
    same = bool((rep.end_errors('free') == job.end_errors('free')).all())
    name = rep.kernel_name()
    rep.release()
    b = 120 * R * n + 72 * R
    return {'name': 'mechanisation_only', 'workload': 'FreeIntegration.run alone on the device-resident sensors of the C2 batch '
            '(%d runs x %d samples; 48 B read + 72 B written per sample*MC)' % (R, n), 'dtype': 'f64',
            'sample_MC_per_s': R * n / avg * 1e3, 'kernel_ms_min': mn,
            'roofline': roofline(b, avg, name, (traffic or {}).get(name, {}).get('hbm_bytes_per_launch')),
            'bit_identical_to_fused_kernel': same, 'placement': brief_placement(placed)}


PMC_CUT_SAMPLES = 8192      # the C3-shaped launches of the --pmc-child workload are cut to this many samples
VIB_LEG = dict(vib_accel={'type': 'random', 'x': 0.294, 'y': 0.294, 'z': 0.294},
               vib_gyro={'type': 'random', 'x': 0.5 * math.pi / 180, 'y': 0.5 * math.pi / 180, 'z': 0.5 * math.pi / 180})


def cut_truth(truth, n):
    return {k: (v[:n] if hasattr(v, 'shape') and getattr(v, 'ndim', 0) >= 1 and v.shape[0] > n else v) for k, v in truth.items()}


def leg_mc(ginsim, workloads, ctx, name, desc, profile, fs, rf, R, keep, precision, reps, gps=False, pmc=None, traffic=None,
           cut=None, valu_too=False, read_bytes_per_unit=0, **job_kw):
    ini, truth, _ = workloads.truth_from_profile(profile, fs, rf, fs_gps=10.0 if gps else 0.0, gps=gps)
    if cut:
        truth = cut_truth(truth, cut)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    n = truth['ref_accel'].shape[0]
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, seed=SEED, keep_sensors=keep, keep_traj=keep,
                               precision=precision, **job_kw)
    job.run()
    placed = job.placement() if keep else None
    avg, mn = time_launches(ctx, job.launch, reps, warm=2 if reps > 2 else 0)
    st = job.stats('free')
    unit = ((BYTES_PER_SAMPLE_MC if precision == 'f64' else BYTES_PER_SAMPLE_MC // 2) if keep else 0) + read_bytes_per_unit
    alg = unit * R * n + 72 * R
    kname = job.kernel_name()
    if keep:
        roof = roofline(alg, avg, kname, (traffic or {}).get(kname, {}).get('hbm_bytes_per_launch'))
        if valu_too:       # a materialising launch that is nevertheless bound by VALU issue: say so with its own object
            v = valu_roofline(pmc, kname, avg, 1.0, 'VALU-busy SIMD cycles of this launch (rocprofv3 --pmc pass of `bench.py --pmc-child`, '
                              'same kernel, same size) against 1024 SIMDs x 2.4 GHz')
            if v is not None:
                roof = dict(v, hbm=roof, note=v['note'] + '; the HBM figure of the same launch is under "hbm"')
    else:
        # nothing is written per sample (72 B per RUN): compute-bound by construction -> the VALU roofline
        roof = valu_roofline(pmc, kname, avg, (n - 1.0) / (PMC_CUT_SAMPLES - 1.0),
                             'stats-only launch: 72 B per RUN reach HBM, the launch is bound by VALU issue; counters from a launch of '
                             'this kernel cut to %d samples (rocprofv3 --pmc pass of `bench.py --pmc-child`)' % PMC_CUT_SAMPLES)
        if roof is None:
            roof = {'bound': 'valu', 'achieved': None, 'peak': SIMDS * SPEC_CLOCK_HZ, 'unit': 'VALU-busy SIMD-cycles/s', 'frac': None,
                    'traffic': None, 'kernel': kname, 'kernel_ms_avg': avg, 'note': 'no PMC pass available in this run (--pmc off or '
                    'rocprofv3 missing): see profiles/*_pmc_counters.csv for this kernel'}
    out = {'name': name, 'workload': desc, 'dtype': precision, 'runs': R, 'samples_per_run': n, 'materialised': keep,
           'sample_MC_per_s': R * n / avg * 1e3, 'kernel_ms_min': mn,
           'roofline': roof,
           'result': {'att_std_deg': (st.std[:3] * 57.29577951308232).tolist(), 'vel_std_mps': st.std[6:9].tolist(), 'runs': st.count}}
    if placed is not None:
        out['placement'] = brief_placement(placed)
    job.release()
    return out


def sensor_generation_roofline(pmc, runs, n, gen_ms, pass_b='ginsim::series_kernel<3>'):
    """The three launches of the time-parallel sensor path write 48 B per sample of a run but are bound by VALU issue (VERDICT r04:
    pass B 0.95 of the chip's VALU issue cycles): bound = valu, from the counters of the two big kernels in the --pmc-child
    pass (each against its own duration there); the HBM figure of the whole generation stays beside it."""
    short = pass_b.replace('ginsim::', '')
    hbm = roofline(48.0 * runs * n, gen_ms, 'series_kernel<0> + series_scan_kernel + ' + short, None,
                   note='48 B written per sample of a run (accel3 + gyro3 doubles); the three launches of the time-parallel path')
    parts = {}
    for k in ('ginsim::series_kernel<0>', pass_b):
        v = valu_roofline(pmc, k, None, 1.0, 'its own duration in the counter pass')
        if v is not None:
            parts[k] = {'frac': v['frac'], 'valu_busy_at_the_clock_it_ran': v['valu_busy_at_the_clock_it_ran'], 'kernel_ms': v['kernel_ms_avg'],
                        'SQ_ACTIVE_INST_VALU': v['counters_per_cut_launch'].get('SQ_ACTIVE_INST_VALU')}
    if not parts:
        return dict(hbm, note=hbm['note'] + '; VALU-bound (profiles/*_pmc_counters.csv), no counter pass in this run')
    t = sum(p['kernel_ms'] for p in parts.values())
    frac = sum(p['frac'] * p['kernel_ms'] for p in parts.values()) / t
    return {'bound': 'valu', 'achieved': frac * SIMDS * SPEC_CLOCK_HZ, 'peak': SIMDS * SPEC_CLOCK_HZ, 'unit': 'VALU-busy SIMD-cycles/s',
            'frac': frac, 'traffic': None, 'kernel': 'series_kernel<0> (pass A) + ' + short + ' (pass B), weighted by their time',
            'kernel_ms_avg': gen_ms, 'per_kernel': parts, 'hbm': hbm,
            'note': 'the generation is bound by VALU issue, not by its stores; the HBM figure of the same launches is under "hbm"'}


def leg_allan(ginsim, workloads, ctx, runs=32, seconds=3600.0, fs=400.0, pmc=None, calls=30, warm=40):
    """BASELINE config 5, second half, END TO END on the device: static truth for 3600 s @ 400 Hz (n = 1 440 000) ->
    accel + gyro series of `runs` runs generated by the time-parallel sensor kernels -> re-layout -> ONE Allan call over all
    6 x runs series (46 averaging factors).  Algorithmic bytes of the Allan call: 8 B per sample (one read of the series)."""
    import numpy as np
    from ginsim._lib import check
    text = open(workloads.profile_path('static_1800s')).read().split('\n')
    ini, _ = workloads.parse_motion('\n'.join(text[:4]))
    seg = np.array([[1.0, 0, 0, 0, 0, 0, 0, seconds, 0.0]])
    raw = ginsim.pathgen(ini, seg, fs, 0.0, workloads.HIGH_MOBILITY, 1)
    truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
             'ref_pos': raw['nav'][:, 1:4], 'ref_vel': raw['nav'][:, 4:7], 'ref_att': raw['nav'][:, 7:10]}
    n = truth['ref_accel'].shape[0]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ctx, fs, 1, truth, acc, gyr, None, runs=runs, algos=(), seed=SEED, keep_sensors=True)
    job.run()
    tau, ad = job.allan(fs)                                        # warm-up (scratch allocation)
    gen_ms, gen_min = time_launches(ctx, job.launch, 10)
    t0 = time.perf_counter()
    tau, ad = job.allan(fs)
    e2e_wall_ms = (time.perf_counter() - t0) * 1e3
    S = 6 * runs
    if job.sensor_layout == 'series' or runs == 1:       # the job's own buffer IS [sensor][run][axis][n]
        tmp, x = None, job.buffer('accel')
    else:
        tmp = x = ctx.malloc(8 * S * n)
        for i, nm in enumerate(('accel', 'gyro')):
            check(ginsim.lib.ginsim_runs_to_series(ctx.handle, job.buffer(nm).ptr, 3, n, runs, tmp.at(i * 24 * n * runs)))
    for _ in range(warm):       # warm-up: the call settles after ~30 of them (WARM_MS)
        ginsim.allan_var(ctx, x, n, S, n, fs)
    ms = []
    for _ in range(calls):      # every call ends with a synchronisation (the sums are on the host): the calls cannot overlap
        ctx.timer_begin()
        ginsim.allan_var(ctx, x, n, S, n, fs)
        ms.append(ctx.timer_end())
    # The same series from ANOTHER allocation of this process: the call's time is bimodal by where a buffer landed (the backing, not
    # the address: profiles/r05_allan_buffer_placement.json), +-4 % on one box.  The roofline below stays the job's own buffer.
    other = None
    if calls >= 10:
        cp = ctx.malloc(8 * S * n + 4096)
        check(ginsim.lib.ginsim_runs_to_series(ctx.handle, x.ptr, 1, S * n, 1, cp.ptr))        # C = 1, R = 1: a plain device copy
        for _ in range(warm // 2):
            ginsim.allan_var(ctx, cp, n, S, n, fs)
        ms2 = []
        for _ in range(calls):
            ctx.timer_begin()
            ginsim.allan_var(ctx, cp, n, S, n, fs)
            ms2.append(ctx.timer_end())
        a2 = sum(ms2) / len(ms2)
        other = {'kernel_ms_avg': a2, 'ms_min': min(ms2), 'frac': 8.0 * S * n / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 'note': 'a device copy of the same series in a buffer allocated for this measurement'}
        cp.free()
    if tmp is not None:
        tmp.free()
    avg = sum(ms) / len(ms)
    # per call: the level kernels + the finishing launch; the sensor generation's own kernels are named series_*
    call_traffic = pmc_traffic_of_call(pmc, 'ginsim::allan_', 'ginsim::allan_tail_kernel')
    # white-noise known answer: AD(tau) = ARW / sqrt(tau) at tau = 1 s
    k1 = int(np.argmin(np.abs(tau - 1.0)))
    arw = float(np.asarray(gyr['arw'])[0])
    out = {'name': 'C5_allan_end_to_end', 'dtype': 'f64',
           'workload': 'static %g s @ %g Hz (n = %d), mid-accuracy IMU, %d runs: sensor generation -> re-layout -> Allan variance '
                       'of %d series, %d averaging factors' % (seconds, fs, n, runs, S, tau.size),
           'sensor_generation_ms': gen_ms, 'sensor_generation_ms_min': gen_min, 'sensor_kernel': job.kernel_name(),
           'sensor_layout': job.sensor_layout,
           'sensor_generation_roofline': sensor_generation_roofline(pmc, runs, n, gen_ms, job.kernel_name()),
           'relayout_plus_allan_wall_ms': e2e_wall_ms, 'allan_call_ms_min': min(ms),
           'samples_per_s_allan_call': S * n / avg * 1e3,
           'roofline': roofline(8.0 * S * n, avg, 'ginsim_allan (every kernel of the call, up to the synchronisation that returns the sums)',
                                (call_traffic or {}).get('hbm_bytes_per_call'), traffic_detail=call_traffic,
                                traffic_over_algorithmic=(call_traffic['hbm_bytes_per_call'] / (8.0 * S * n)) if call_traffic else None),
           'result': {'ad_gyro_x_at_1s_over_arw': float(ad['gyro'][:, k1, 0].mean() / arw * np.sqrt(tau[k1]))}}
    if other is not None:
        out['same_series_in_another_allocation'] = other
    job.release()
    return out


def leg_sim_e2e(workloads):
    """End-to-end wall of the drop-in Sim ("also report end-to-end Sim.run wall"): constructor + run() + results()
    as demo_free_integration.py calls them, host work included (motion parsing, native pathgen, uploads, the launch, the
    device reductions, the summary text) -- C2 with everything kept on the device, and C3 (statistics only, pathgen of
    193 036 samples on the host inside the wall)."""
    import contextlib
    import io
    import numpy as np
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    out = {'name': 'sim_e2e', 'dtype': 'f64', 'workload': 'Sim(...).run(R); Sim.results() through the drop-in package, wall clock'}
    for tag, profile, fs, fs_gps, rf, R, axis, gps in (('C2', 'turn_90deg', 100.0, 0.0, 1, 65536, 6, False),
                                                      ('C3', 'long_drive', 200.0, 10.0, 0, 262144, 9, True)):
        csv = workloads.profile_path(profile)
        ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
        ini[0:2] *= np.pi / 180
        ini[6:9] *= np.pi / 180
        best, walls = None, []
        for rep in range(4):                         # the first pass pays the device allocations; later ones find them in the context's pool
            imu = imu_model.IMU(accuracy='mid-accuracy', axis=axis, gps=gps)
            t0 = time.perf_counter()
            # C3 as BASELINE configs[2] names it: 9-axis IMU + GPS error model.  The geomagnetic field at the start is an INPUT
            # (the WMM model is outside the path: a fixed vector here); keep_runs=2 materialises accel / gyro / mag / GPS series and
            # trajectories of two runs next to the statistics over all of them (the full series would be 6 TB)
            extra = dict(geo_mag_n=[33.0, -2.4, 36.5], keep_runs=2) if tag == 'C3' else {}
            sim = ins_sim.Sim([fs, fs_gps, fs if axis == 9 else 0.0], csv, ref_frame=rf, imu=imu, mode=None, env=None,
                              algorithm=free_integration.FreeIntegration(ini), seed=SEED, **extra)
            sim.run(R)
            t1 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                sim.results(err_stats_start=-1 if tag == 'C2' else 0)
            t2 = time.perf_counter()
            n = int(sim.dmgr.time.data.shape[0])
            rec = {'runs': R, 'samples_per_run': n, 'run_wall_s': t1 - t0, 'results_wall_s': t2 - t1,
                   'sample_MC_per_s_end_to_end': R * n / (t2 - t0),
                   'statistics': 'end point' if tag == 'C2' else 'process error of every run from t = 0 (the reference default), accumulated online',
                   'imu': '%d-axis%s' % (axis, ' + GPS' if gps else ''),
                   'kept_runs': sorted(int(k) for k in sim.dmgr.accel.data.keys()) if tag == 'C3' else 'all',
                   'series_available': [k for k in ('accel', 'gyro', 'mag', 'gps', 'pos') if k in sim.dmgr.available]}
            best = rec if best is None or rec['sample_MC_per_s_end_to_end'] > best['sample_MC_per_s_end_to_end'] else best
            walls.append(t2 - t0)
            del sim
        best['wall_s_every_construction'] = walls       # [0] includes hipMalloc of the series buffers (1 .. 280 ms for 3-5 GB)
        out[tag] = best
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


# --------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--runs-per-gpu', type=int, default=0, help='default: 65 536 at N = 1 (C2), 131 072 at N > 1 (C4)')
    ap.add_argument('--profile', default='turn_90deg')
    ap.add_argument('--fs', type=float, default=100.0)
    ap.add_argument('--ref-frame', type=int, default=1)
    ap.add_argument('--stats-only', action='store_true', help='do not materialise sensors/trajectories')
    ap.add_argument('--precision', choices=['f64', 'f32'], default='f64', help="f32 = BASELINE config 5's single-precision kernel")
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0, help='0 disables the CPU baseline leg')
    ap.add_argument('--no-legs', action='store_true', help='skip the configs[] legs (C3, C4 share, C5, Allan, mechanisation)')
    ap.add_argument('--placement', choices=['placed', 'asis'], default='placed',
                    help="placed (the library's default, what an unconfigured Sim gets): the materialised series are carved from the "
                         "device's placed arena (ABI 7: stripes cycling through the three classes of physical memory); asis: plain "
                         "hipMalloc, wherever the driver puts the planes.  Either way the line carries BOTH rooflines (N = 1): "
                         "`roofline` for --placement, `roofline_asis` / `roofline_placed` for the other one, same process")
    ap.add_argument('--no-repeat', action='store_true', help='N = 1: do not time the K steps a second time (headline_again)')
    ap.add_argument('--pmc', choices=['live', 'file', 'off'], default='live',
                    help='roofline.traffic: rocprofv3 PMC passes of this build (live), the stamped profiles/pmc_traffic.json, or null')
    ap.add_argument('--pmc-child', action='store_true', help='internal: the short workload the live PMC passes profile')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only for tests)')
    ap.add_argument('--exchange', choices=['abi', 'torch'], default=None,
                    help="N > 1: 'abi' = the library's own RCCL all-gather on the kernel stream (ginsim_end_stats_all_begin/_finish, "
                         "default with --backend nccl), 'torch' = torch.distributed all-reduce of the record table (default otherwise)")
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise torch.distributed and run the exchange also with ONE rank (a one-GPU box can then execute '
                         'the RCCL code path: communicator set-up, device-tensor all-reduce, non-blocking work handles)')
    ap.add_argument('--shared-device', action='store_true',
                    help='TEST ONLY: every rank uses GPU 0 (exercises the N > 1 control flow on a one-GPU box)')
    args = ap.parse_args()
    if args.pmc_child:
        args.steps, args.warmup, args.cpu_baseline_seconds, args.no_legs, args.pmc, args.no_repeat = 3, 1, 0.0, True, 'off', True

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # started without a launcher: become `torch.distributed.run --nproc-per-node N bench.py <same arguments>` (one rank per GPU)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
               '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import ginsim
    from ginsim import workloads, distributed

    if args.shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    if args.shared_device and world > 1:
        # TEST ONLY: every rank builds its arena on the ONE GPU they share -- each search may hold 1 / (2 N) of the device at
        # most, so that the N > 1 control flow (arena search, then the collectives) runs on a one-GPU box without exhausting it
        os.environ.setdefault('GINSIM_PLACED_BUDGET_GIB', '%d' % max(8, 288 // (2 * world)))
        os.environ.setdefault('GINSIM_PLACED_LIMIT_GIB', '20')
    ctx = ginsim.Context(local_rank)
    if args.placement == 'asis':
        ctx.placed_enabled, ctx.placed_note = False, '--placement asis'

    fs, rf = args.fs, args.ref_frame
    R = args.runs_per_gpu or (65536 if world == 1 else 131072)
    ini, truth, _ = workloads.truth_from_profile(args.profile, fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    n = truth['ref_accel'].shape[0]
    keep = not args.stats_only
    job = None
    for turn in range(world if (args.shared_device and world > 1) else 1):
        # (TEST ONLY, --shared-device: the ranks build their arenas on the one GPU in turn -- two searches at once on ONE device
        # disturb each other's probe timings; with a GPU per rank every rank searches its own at the same time)
        if not (args.shared_device and world > 1) or turn == rank:
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=SEED,
                                       keep_sensors=keep, keep_traj=keep, precision=args.precision)
        if args.shared_device and world > 1:
            dist.barrier()
    unit_bytes = BYTES_PER_SAMPLE_MC if args.precision == 'f64' else BYTES_PER_SAMPLE_MC // 2
    placement = job.placement() if keep else None           # the arena was built (one search) inside the constructor: set-up
    group = dist.group.WORLD if use_dist else None
    device = torch.device('cuda', local_rank) if args.backend == 'nccl' else torch.device('cpu')
    nsteps = args.warmup + args.steps
    # HIP events bracket the MC kernel of every `stride`-th step (the context has 8192 event slots)
    stride = max(1, -(-2 * nsteps // 8192))

    # Two batches are in flight behind the one being integrated: batch s-1's on-device reduction (its 28-double record
    # lands in a pinned slot) and batch s-2's all-reduce (issued non-blocking one step earlier).  The host therefore
    # never waits for a collective that has not had a whole kernel time to complete -- also when the RCCL kernel
    # cannot be scheduled next to the MC kernel, which fills every CU's LDS.
    pending_stats, pending_coll = [], []
    exchange, exchange_note = (args.exchange or ('abi' if args.backend == 'nccl' else 'torch')) if use_dist else None, None
    if exchange == 'abi':
        # the library's own communicator (RCCL, resolved at run time); torch.distributed only carries the 128-byte id.  All ranks
        # must agree on the outcome, so the success flags are reduced before anyone relies on it.
        ok = 1.0
        try:
            distributed.init_abi_comm(ctx, group, device)
        except Exception as e:                                     # noqa: BLE001 -- fall back to the torch exchange, say so
            ok, exchange_note = 0.0, 'abi exchange unavailable on rank %d: %s' % (rank, repr(e)[:160])
        flag = torch.tensor([ok], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag.item()) < 1.0:
            if ok:
                ctx.comm_destroy()
            exchange, exchange_note = 'torch', exchange_note or 'abi exchange unavailable on another rank'
    # what the communicators THEMSELVES say (ncclCommCount / UserRank / CuDevice), from every rank: the line's proof that RCCL saw N ranks
    rccl_seen = None
    if use_dist:
        mine = None
        if exchange == 'abi':
            try:
                mine = dict(zip(('ranks', 'rank', 'device'), ctx.comm_query()))
            except Exception as e:                                 # noqa: BLE001
                mine = {'error': repr(e)[:120]}
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        if rank == 0 and any(r is not None for r in rows):
            counts = [r.get('ranks') for r in rows if r and 'ranks' in r]
            rccl_seen = {'rccl_ranks': min(counts) if counts and len(counts) == world else None, 'every_rank': rows,
                         'note': 'ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the library\'s own communicator on every rank '
                                 '(ginsim_comm_query); rccl_ranks = the smallest count, None unless every rank answered'}

    def collect():
        """finish what can be finished: the all-reduce of batch s-2, then the record of batch s-1 -> issue its all-reduce"""
        merged = distributed.allreduce_stats_end(pending_coll.pop()) if pending_coll else None
        if pending_stats:
            part = job.stats_finish(pending_stats.pop())
            pending_coll.append(distributed.allreduce_stats_begin(part, group, device))
        return merged

    def drain():
        merged = None
        if exchange == 'abi':
            while pending_stats:
                merged = job.stats_all_finish(pending_stats.pop())
            return merged
        while pending_stats or pending_coll:
            m = collect()
            merged = m if m is not None else merged
        return merged

    launched = [0]                      # launches of the headline kernel by this process so far (the kernel trace's index of a launch)
    _launch = job.launch

    def counted_launch():
        launched[0] += 1
        _launch()
    job.launch = counted_launch

    def step(s):
        job.params.run_offset = (s * world + rank) * R      # a fresh batch of global run ids every step
        if s % stride == 0:
            ctx.event_record(2 * (s // stride))
        job.launch()
        if s % stride == 0:
            ctx.event_record(2 * (s // stride) + 1)
        if exchange == 'abi':       # reduction -> RCCL all-gather -> pinned copy, all on the kernel stream; one batch behind
            merged = job.stats_all_finish(pending_stats.pop()) if pending_stats else None
            job.stats_all_begin('free', s & 1)
            pending_stats.append(s & 1)
            return merged
        merged = collect()
        job.stats_begin('free', s & 1)
        pending_stats.append(s & 1)
        return merged

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    # pre-warm by TIME: WARM_MS of this kernel's own launches, outside the counted --warmup steps (whatever the driver passes
    # for --warmup, the timed steps then run in the steady state the profiles are taken in)
    prewarm_ms, done = 0.0, 0
    while prewarm_ms < WARM_MS and done < 64 and not args.pmc_child:
        ctx.timer_begin()
        job.launch()
        prewarm_ms += ctx.timer_end()
        done += 1

    # N > 1: the same per-GPU load on ONE GPU at a time while the other ranks wait -- the reference point scaling efficiency
    # needs, from EVERY rank (a slow GPU shows up here before it shows up as the step time of the whole job)
    single = None
    if use_dist and world > 1:
        mine = None
        for turn in range(world):
            fence()
            if rank == turn:
                k = max(2, min(args.steps, 20))
                ctx.sync()
                t0 = time.perf_counter()
                for s in range(k):
                    job.params.run_offset = s * R
                    job.launch()
                    job.stats_begin('free', s & 1)
                    job.stats_finish(s & 1)
                ctx.sync()
                dt = time.perf_counter() - t0
                mine = {'rank': rank, 'value': R * n * k / dt, 'ms_per_step': dt / k * 1e3, 'steps': k}
        fence()
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if rank == 0:
            vals = [e['value'] for e in everyone]
            single = {'value': everyone[0]['value'], 'unit': 'sample*MC/s', 'ms_per_step': everyone[0]['ms_per_step'],
                      'steps': everyone[0]['steps'], 'runs': R,
                      'every_rank': {'value': vals, 'min': min(vals), 'max': max(vals), 'argmin': int(np.argmin(vals))},
                      'note': 'each rank alone in turn (the other ranks idle at a barrier), same runs per GPU, launch + device '
                              'reduction per step, no exchange: efficiency(N) = value / (N x this); `value` is rank 0, `every_rank` '
                              'lists all of them'}

    if exchange == 'abi':
        # one batch through BOTH exchanges before anything is timed: the library's all-gather must give the record the
        # torch.distributed all-reduce gives; if not, the timed region uses the torch exchange and the line says so
        job.params.run_offset = rank * R
        job.launch()
        job.stats_all_begin('free', 0)
        a = job.stats_all_finish(0)
        b = distributed.allreduce_stats(job.stats('free'), group, device)
        same = a.count == b.count and np.allclose(a.mean, b.mean, rtol=1e-12, atol=1e-18) and np.allclose(a.m2, b.m2, rtol=1e-12) \
            and np.array_equal(a.maxabs, b.maxabs)
        flag = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag.item()) < 1.0:
            ctx.comm_destroy()
            exchange, exchange_note = 'torch', 'abi exchange disagreed with the torch.distributed all-reduce on the check batch'

    for s in range(args.warmup):
        step(s)
    drain()
    fence()
    timed_first = launched[0]           # the timed steps are launches timed_first .. timed_first + steps - 1 of this kernel in this process
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        step(s)
    merged = drain()                                        # the last batches' exchanges are inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = [ctx.event_elapsed(2 * (s // stride), 2 * (s // stride) + 1) for s in range(args.warmup, nsteps) if s % stride == 0]
    kern_avg_ms = float(np.mean(kern_ms))

    # The same timed region ONCE MORE in this process, a few seconds later (N = 1): how much of the run-to-run spread of this
    # store-bound kernel is the box and how much is the moment.  `value` stays the first region (the K steps the contract names).
    again = None
    if world == 1 and not args.pmc_child and not args.no_repeat:
        time.sleep(3.0)
        spent, done = 0.0, 0
        while spent < WARM_MS and done < 64:
            ctx.timer_begin()
            job.launch()
            spent += ctx.timer_end()
            done += 1
        fence()
        t1 = time.perf_counter()
        for s in range(args.warmup, nsteps):
            step(s)
        merged2 = drain()
        fence()
        dt = time.perf_counter() - t1
        k2 = [ctx.event_elapsed(2 * (s // stride), 2 * (s // stride) + 1) for s in range(args.warmup, nsteps) if s % stride == 0]
        again = {'value': float(R) * n * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3, 'kernel_ms_avg': float(np.mean(k2)),
                 'seconds_after_the_first': t1 - t0 - elapsed, 'same_statistics': bool(merged2.count == merged.count and
                                                                                      np.array_equal(merged2.m2, merged.m2)),
                 'note': 'the K timed steps repeated in the same process after a 3 s pause and the same time-based pre-warm'}
    # The SAME launches with the other placement, same process (N = 1): plain hipMalloc planes when the headline ran placed, placed
    # planes when it ran --placement asis.  K launches back to back after the same time-based pre-warm, HIP events around each.
    other = None
    if world == 1 and keep and not args.pmc_child and not args.shared_device:
        want_placed = args.placement == 'asis'
        saved = (ctx.placed_enabled, ctx.placed_note)
        if want_placed:
            ctx.placed_enabled = os.environ.get('GINSIM_PLACED', '1') != '0'
        try:
            j2 = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=SEED, keep_sensors=True,
                                      keep_traj=True, precision=args.precision, placed=want_placed)
            k = max(2, min(args.steps, 100))
            avg2, mn2 = time_launches(ctx, j2.launch, k)
            other = {'placement': 'placed' if want_placed else 'asis', 'kernel_ms_avg': avg2, 'kernel_ms_min': mn2, 'launches': k,
                     'got': brief_placement(j2.placement())}
            j2.release()
        except ginsim.GinsimError as e:                           # e.g. not enough memory for a second set of planes
            other = {'placement': 'placed' if want_placed else 'asis', 'error': str(e)[:200]}
        ctx.placed_enabled, ctx.placed_note = saved
    per_rank = None
    if use_dist and world > 1:          # the slowest GPU sets the step time: make it visible in the line
        rows = [None] * world
        unit_b = (BYTES_PER_SAMPLE_MC if args.precision == 'f64' else BYTES_PER_SAMPLE_MC // 2) if keep else 0
        dist.all_gather_object(rows, {'rank': rank, 'kernel_ms_avg': kern_avg_ms, 'kernel_ms_max': float(np.max(kern_ms)),
                                      'device': ctx.name(), 'placement': brief_placement(placement)})
        ks = [r['kernel_ms_avg'] for r in rows]
        per_rank = {'kernel_ms_avg': ks, 'min': min(ks), 'max': max(ks), 'argmax': int(np.argmax(ks)),
                    'kernel_ms_max': [r['kernel_ms_max'] for r in rows],
                    # every rank's own roofline: algorithmic bytes of ITS launch over ITS kernel time in the timed region
                    'roofline_frac': [(unit_b * R * n + 72 * R) / (k * 1e-3) / 1e9 / HBM_PEAK_GBS for k in ks],
                    'placement': [r['placement'] for r in rows]}
    assert merged.count == world * R, (merged.count, world * R)

    if args.pmc_child:      # every kernel a roofline object of the line is about, in the same passes
        if keep and args.precision == 'f64':
            leg_mechanisation(ginsim, ctx, job, fs, rf, truth, ini, R, n, None, reps=3)
            job.release()
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=SEED, keep_sensors=True,
                                       keep_traj=True, precision='f32')
            for _ in range(4):
                job.launch()
            ctx.sync()
            job.release()
            job = None
            # C3's kernels (compute-bound: VALU counters), cut to PMC_CUT_SAMPLES samples
            leg_mc(ginsim, workloads, ctx, 'c3', '', 'long_drive', 200.0, 0, 262144, False, 'f64', 2, gps=True, cut=PMC_CUT_SAMPLES,
                   proc_first=0, end_ned=True)
            leg_mc(ginsim, workloads, ctx, 'c3e', '', 'long_drive', 200.0, 0, 262144, False, 'f64', 2, gps=True, cut=PMC_CUT_SAMPLES)
            leg_allan(ginsim, workloads, ctx, calls=3, warm=1)       # config 5: sensor generation + three Allan calls
            leg_mc(ginsim, workloads, ctx, 'vib', '', 'turn_90deg', 100.0, 1, 65536, True, 'f64', 2, **VIB_LEG)     # the vibration leg's kernel
        if job is not None:
            job.release()
        ctx.close()
        return

    if rank == 0:
        total_units = float(world) * R * n * args.steps
        alg_bytes = (unit_bytes if keep else 0) * R * n + 72 * R      # per launch, per GPU
        r2d = 180.0 / np.pi
        build = lib_hash()
        kname = job.kernel_name()
        given_name = 'ginsim::mc_kernel<%d, 1, true, false, 0, false>' % rf
        traffic, traffic_source, pmc = None, None, None
        if world == 1 and args.precision == 'f64' and keep:
            if args.pmc == 'live':
                pmc = pmc_live()
                traffic = pmc_traffic(pmc, [k for k in (pmc or {}) if 'mc_kernel' in k])
                traffic_source = 'live: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of `bench.py --pmc-child` on this build'
            if traffic is None and args.pmc in ('live', 'file'):
                traffic, traffic_source = pmc_traffic_file(build), 'profiles/pmc_traffic.json (same libginsim.so hash)'
            if traffic is None:
                traffic_source = None
        cfg_name = ('BASELINE configs[1] (C2)' if (world == 1 and R == 65536) else
                    'BASELINE configs[3] (C4: %d runs over %d GPUs)' % (world * R, world) if R == 131072 else 'custom')
        out = {
            'metric': 'Monte-Carlo IMU samples integrated/sec', 'value': total_units / elapsed,
            'unit': 'sample*MC/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': '%s: %s @%g Hz (n=%d), mid-accuracy 6-axis IMU, ref_frame=%d, free_integration, '
                                   '%d MC runs per GPU, %s' %
                                   (cfg_name, args.profile, fs, n, rf, R,
                                    'sensors+trajectories materialised (%d B/sample*MC)' % unit_bytes if keep else 'stats-only'),
                       'runs_per_gpu': R, 'samples_per_run': n, 'total_runs_per_step': world * R,
                       'parallelism': ('mc-shard x%d, one exchange of the 28-double stats record per step: %s' % (world, (
                           'RCCL all-gather behind the C ABI on the kernel stream (ginsim_end_stats_all_begin/_finish)' if exchange == 'abi'
                           else 'torch.distributed all-reduce (%s)' % ('RCCL' if args.backend == 'nccl' else args.backend))
                           + ('; ' + exchange_note if exchange_note else ''))) if use_dist else
                                      'one GPU, no collective (runs shard over ranks by global run id at N > 1)',
                       'device': ctx.name(), 'libginsim_sha256': build,
                       'rng': 'Philox4x32-7, 3 blocks per IMU step, one word per normal by piecewise-cubic inversion defined bit-exactly '
                              'in single precision (|z| <= 6.23)'},
            'prewarm_ms': prewarm_ms,
            'roofline': roofline(alg_bytes, kern_avg_ms, kname, (traffic or {}).get(kname, {}).get('hbm_bytes_per_launch'),
                                 traffic_source=traffic_source,
                                 timed_launches={'first': timed_first, 'count': args.steps,
                                                 'note': 'zero-based index among the launches of this kernel at this grid size in the '
                                                         'process: the rows of a rocprofv3 kernel trace of the same command to average'},
                                 note='writes exactly its algorithmic bytes; store-bound: a pure non-temporal fill of the same 15-plane '
                                      'pattern takes ~1.15 ms, see DESIGN.md section 4.1'),
            'result': {'att_std_deg': (merged.std[:3] * r2d).tolist(), 'vel_std_mps': merged.std[6:9].tolist(),
                       'runs': merged.count},
        }
        # where the planes lie -- top-level keys, so that the driver's record keeps them
        out['placement'] = {'mode': args.placement,
                            'job': placement,
                            'note': 'placed = the library default (what an unconfigured Sim gets): the 15 output planes are carved from the '
                                    "device's placed arena, whose 512 MiB stripes cycle through the three classes of physical memory "
                                    '(ABI 7, csrc/placed.hip); arena.search_seconds / chunks_created / probes are the one-off cost of '
                                    'building it, before the warm-up'}
        if other is not None and 'kernel_ms_avg' in other:
            key = 'roofline_' + other['placement']
            out[key] = roofline(alg_bytes, other['kernel_ms_avg'], kname, None, kernel_ms_min=other['kernel_ms_min'],
                                launches=other['launches'], got=other['got'],
                                note='the same launch in the same process with the OTHER placement (%s), back to back after the '
                                     'timed region' % ('plain hipMalloc planes' if other['placement'] == 'asis' else 'planes from the placed arena'))
        elif other is not None:
            out['roofline_' + other['placement']] = other
        if again is not None:
            out['headline_again'] = again
            out['roofline']['frac_again'] = alg_bytes / (again['kernel_ms_avg'] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if world > 1:
            out['cpu_baseline_note'] = 'omitted at N > 1: the CPU baseline is timed on rank 0 at N = 1 only (bench contract)'
        if single is not None:
            out['per_gpu_single'] = single
            out['efficiency'] = out['value'] / (world * single['every_rank']['min'])
            out['efficiency_note'] = 'value / (N x the slowest rank\'s single-GPU rate on the same per-GPU load): what this line says about scaling; the driver computes its own from the per-N values'
        if rccl_seen is not None:
            out.update(rccl_seen)
        if per_rank is not None:
            out['per_rank'] = per_rank
        if world == 1 and not args.no_legs:
            legs = []

            if keep and args.precision == 'f64':
                legs.append(leg_mechanisation(ginsim, ctx, job, fs, rf, truth, ini, R, n, traffic))
                out['mechanisation_only'] = legs[-1]
            job.release()
            job = None
            legs.append(leg_mc(ginsim, workloads, ctx, 'C4_per_gpu_share', 'BASELINE configs[3] per-GPU share: turn_90deg @100 Hz, '
                               '131 072 runs, fp64, materialised', 'turn_90deg', 100.0, 1, 131072, True, 'f64', 10))
            c3 = dict(pmc=pmc)
            legs.append(leg_mc(ginsim, workloads, ctx, 'C3', 'BASELINE configs[2]: long_drive @200 Hz (n = 193 036), ref_frame 0, 262 144 '
                               'runs, fp64; trajectories would be 6 TB, so the kernel accumulates the per-run process-error statistics '
                               '(what Sim.results() prints by default) and both end-point records online; GPS / magnetometer series '
                               'are generated for the kept subset only (Sim keep_runs), they do not enter the integration',
                               'long_drive', 200.0, 0, 262144, False, 'f64', 2, gps=True, proc_first=0, end_ned=True, **c3))
            legs.append(leg_mc(ginsim, workloads, ctx, 'C3_end_point_only', 'the same launch with end-point statistics only (r01 form)',
                               'long_drive', 200.0, 0, 262144, False, 'f64', 2, gps=True, **c3))
            legs.append(leg_mc(ginsim, workloads, ctx, 'C5_fp32', 'BASELINE configs[4]: fp32 kernel on the C2 workload, 65 536 runs, '
                               'materialised (60 B/sample*MC)', 'turn_90deg', 100.0, 1, 65536, True, 'f32', 20, traffic=traffic))
            legs.append(leg_mc(ginsim, workloads, ctx, 'C5_fp32_262144', 'fp32 kernel, 262 144 runs, materialised', 'turn_90deg', 100.0, 1,
                               262144, True, 'f32', 10))
            # Sim(env=...) (beyond BASELINE's configurations, all of which use env=None): the C2 launch in a vibration environment
            legs.append(leg_mc(ginsim, workloads, ctx, 'C2_vibration_random', 'the C2 launch with Sim(env={acc: [0.03 0.03 0.03]g-random, '
                               'gyro: [0.5 0.5 0.5]d-random}): the vibration variant of the wave-specialised kernel (round 5; round 4: the plain kernel, one wavefront per SIMD)', 'turn_90deg',
                               100.0, 1, 65536, True, 'f64', 10,
                               pmc=pmc, valu_too=True, **VIB_LEG))
            # Sim(env=<(n, 4) PSD array>) (ABI 8): the series of both sensors are made before the launch (ginsim_vib_psd_series) and
            # read by it: 120 B written + 48 B read per sample*MC
            import numpy as np
            psd_f = np.array([0.0, 8.0, 11.0, 13.0, 16.0, 50.0])
            psd = lambda u: {'type': 'psd', 'freq': psd_f, 'x': u * np.array([1e-4, 1e-4, 2e-2, 2e-2, 1e-4, 1e-4]),
                             'y': u * np.full(6, 1e-3), 'z': u * np.full(6, 2e-3)}
            legs.append(leg_mc(ginsim, workloads, ctx, 'C2_vibration_psd', 'the C2 launch with Sim(env={acc: PSD array, gyro: PSD array}): '
                               'the vibration variant of the plain kernel reading the six series planes made by ginsim_vib_psd_series '
                               '(120 B written + 48 B read per sample*MC)', 'turn_90deg', 100.0, 1, 65536, True, 'f64', 10,
                               read_bytes_per_unit=48, vib_accel=psd(1.0), vib_gyro=psd(1e-4)))
            legs.append(leg_allan(ginsim, workloads, ctx, pmc=pmc))
            legs.append(leg_sim_e2e(workloads))
            out['configs'] = legs
        if world == 1 and args.cpu_baseline_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(fs, rf, ini, truth, acc, gyr, args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)

    if job is not None:
        job.release()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
