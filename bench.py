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
Workload = BASELINE.json configs[1]: 90-degree-turn profile @100 Hz (n = 1000), 'mid-accuracy' 6-axis IMU,
ref_frame = 1, 65 536 runs per GPU, fp64.  Weak scaling: every rank integrates its own 65 536 runs
(global run ids are disjoint, the Philox counter carries the global id).

Prints ONE JSON line (rank 0).  Inputs (truth, parameters) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_SAMPLE_MC = 120   # accel3 + gyro3 + att3 + pos3 + vel3 doubles, all writes


def cpu_baseline(fs, rf, ini, truth, acc, gyr, seed, budget_s):
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
    c_oracle.mc_run(seed, 0, 64 * cores, fs, rf, truth, acc, gyr, ini)
    probe = time.perf_counter() - t0
    runs = int(max(64 * cores, min(2_000_000, budget_s / max(probe, 1e-6) * 64 * cores)))
    t0 = time.perf_counter()
    c_oracle.mc_run(seed, 0, runs, fs, rf, truth, acc, gyr, ini)
    dt = time.perf_counter() - t0
    return {'value': runs * n / dt, 'unit': 'sample*MC/s', 'cores': cores, 'kind': 'port',
            'sample': '%d runs x %d samples of the same workload through oracle/c/ginsim_oracle.c '
                      '(OpenMP over runs, %.1f s)' % (runs, n, dt)}


def mechanisation_only(ginsim, ctx, job, fs, rf, truth, ini, R, n, reps=10):
    """Outside the timed region: FreeIntegration.run alone for the whole batch -- the given-sensors kernel reads the
    accel/gyro series the last step materialised (48 B) and writes att/pos/vel (72 B) per sample*MC.  This is the
    HBM-bound piece of the path; the fused kernel above is fp64-VALU-bound."""
    rep = ginsim.MonteCarloJob(ctx, fs, rf, truth, None, None, ini, runs=R, keep_traj=True,
                               given={'gyro': job.buffer('gyro'), 'accel': job.buffer('accel')})
    rep.run()
    ms = []
    for _ in range(reps):
        ctx.timer_begin()
        rep.launch()
        ms.append(ctx.timer_end())
    same = bool((rep.end_errors('free') == job.end_errors('free')).all())
    name = rep.kernel_name()
    rep.release()
    avg = sum(ms) / len(ms)
    b = 120 * R * n + 72 * R
    return {'kernel': name, 'kernel_ms_avg': avg, 'algorithmic_bytes_per_launch': b, 'achieved': b / avg / 1e6, 'unit': 'GB/s',
            'peak': HBM_PEAK_GBS, 'frac': b / avg / 1e6 / HBM_PEAK_GBS, 'traffic': pmc_traffic('mc_kernel_rf%d_free_given' % rf),
            'sample_MC_per_s': R * n / avg * 1e3,
            'bit_identical_to_fused_kernel': same}


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or None."""
    path = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(kernel_key, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--runs-per-gpu', type=int, default=65536)
    ap.add_argument('--profile', default='turn_90deg')
    ap.add_argument('--fs', type=float, default=100.0)
    ap.add_argument('--ref-frame', type=int, default=1)
    ap.add_argument('--stats-only', action='store_true', help='do not materialise sensors/trajectories')
    ap.add_argument('--precision', choices=['f64', 'f32'], default='f64', help="f32 = BASELINE config 5's single-precision kernel")
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0, help='0 disables the CPU baseline leg')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only for tests)')
    ap.add_argument('--shared-device', action='store_true',
                    help='TEST ONLY: every rank uses GPU 0 (exercises the N > 1 control flow on a one-GPU box)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import ginsim
    from ginsim import workloads, distributed

    if args.shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    ctx = ginsim.Context(local_rank)

    fs, rf, R, seed = args.fs, args.ref_frame, args.runs_per_gpu, 20260923
    ini, truth, _ = workloads.truth_from_profile(args.profile, fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    n = truth['ref_accel'].shape[0]
    keep = not args.stats_only
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, algos=('free',), seed=seed,
                               keep_sensors=keep, keep_traj=keep, precision=args.precision)
    unit_bytes = BYTES_PER_SAMPLE_MC if args.precision == 'f64' else BYTES_PER_SAMPLE_MC // 2
    group = dist.group.WORLD if world > 1 else None
    device = torch.device('cuda', local_rank) if args.backend == 'nccl' else torch.device('cpu')
    nsteps = args.warmup + args.steps
    # HIP events bracket the MC kernel of every `stride`-th step (the context has 8192 event slots)
    stride = max(1, -(-2 * nsteps // 8192))

    # Two batches are in flight behind the one being integrated: batch s-1's on-device reduction (its 28-double record
    # lands in a pinned slot) and batch s-2's all-reduce (issued non-blocking one step earlier).  The host therefore
    # never waits for a collective that has not had a whole kernel time to complete -- also when the RCCL kernel
    # cannot be scheduled next to the MC kernel, which fills every CU's LDS.
    pending_stats, pending_coll = [], []

    def collect():
        """finish what can be finished: the all-reduce of batch s-2, then the record of batch s-1 -> issue its all-reduce"""
        merged = distributed.allreduce_stats_end(pending_coll.pop()) if pending_coll else None
        if pending_stats:
            part = job.stats_finish(pending_stats.pop())
            pending_coll.append(distributed.allreduce_stats_begin(part, group, device))
        return merged

    def drain():
        merged = None
        while pending_stats or pending_coll:
            m = collect()
            merged = m if m is not None else merged
        return merged

    def step(s):
        job.params.run_offset = (s * world + rank) * R      # a fresh batch of global run ids every step
        if s % stride == 0:
            ctx.event_record(2 * (s // stride))
        job.launch()
        if s % stride == 0:
            ctx.event_record(2 * (s // stride) + 1)
        merged = collect()
        job.stats_begin('free', s & 1)
        pending_stats.append(s & 1)
        return merged

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    drain()
    fence()
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        step(s)
    merged = drain()                                        # the last batches' exchanges are inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = [ctx.event_elapsed(2 * (s // stride), 2 * (s // stride) + 1) for s in range(args.warmup, nsteps) if s % stride == 0]
    kern_avg_ms = float(np.mean(kern_ms))
    assert merged.count == world * R, (merged.count, world * R)

    if rank == 0:
        total_units = float(world) * R * n * args.steps
        alg_bytes = (unit_bytes if keep else 0) * R * n + 72 * R      # per launch, per GPU
        achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
        r2d = 180.0 / np.pi
        out = {
            'metric': 'Monte-Carlo IMU samples integrated/sec', 'value': total_units / elapsed,
            'unit': 'sample*MC/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: %s @%g Hz (n=%d), mid-accuracy 6-axis IMU, ref_frame=%d, '
                                   'free_integration, %d MC runs per GPU, %s' %
                                   (args.profile, fs, n, rf, R,
                                    'sensors+trajectories materialised (%d B/sample*MC)' % unit_bytes if keep else 'stats-only'),
                       'runs_per_gpu': R, 'samples_per_run': n, 'total_runs_per_step': world * R,
                       'parallelism': 'mc-shard x%d, one all-reduce of the 28-double stats record' % world,
                       'device': ctx.name()},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc_traffic('mc_kernel_rf%d_free_%s' % (rf, 'keep' if keep else 'stats')) if args.precision == 'f64' else None,
                         'kernel': job.kernel_name(), 'kernel_ms_avg': kern_avg_ms,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'note': 'fp64 transcendental/VALU-bound, not HBM-bound: see DESIGN.md (roofline)'},
            'result': {'att_std_deg': (merged.std[:3] * r2d).tolist(), 'vel_std_mps': merged.std[6:9].tolist(),
                       'runs': merged.count},
        }
        if world == 1 and keep and args.precision == 'f64':
            out['mechanisation_only'] = mechanisation_only(ginsim, ctx, job, fs, rf, truth, ini, R, n)
        if world == 1 and args.cpu_baseline_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(fs, rf, ini, truth, acc, gyr, seed, args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)

    job.release()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
