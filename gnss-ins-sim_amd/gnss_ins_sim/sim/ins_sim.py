"""INS simulation driver, interface-compatible with the reference's ``Sim``
(gnss_ins_sim/sim/ins_sim.py:27-832):

    Sim(fs, motion_def, ref_frame=0, imu=None, mode=None, env=None, algorithm=None)
    .run(num_times)  .results(data_dir, err_stats_start, gen_kml, extra_opt)  .get_data(names)
    .dmgr  .amgr  .sim_count  .sum

What differs is HOW ``run`` executes (ins_sim.py:164-192, 490-506 in the reference are two serial Python
loops over runs): here the truth comes from the native ``ginsim_pathgen``, and ALL Monte-Carlo runs go
through ONE launch of the fused HIP kernel (noise injection + mechanisation + end-point error) per GPU.
Sensor series and algorithm outputs stay in HBM; ``dmgr.<series>.data`` are mapping views that pull a run
to the host only when it is indexed.  With ``torch.distributed`` initialised (one process per GPU) the runs
are sharded across ranks and the end-point statistics are combined with one all-reduce.

Keyword-only extras (defaults keep the reference behaviour):
    seed               64-bit Philox key.  None: drawn from ``np.random`` (so ``np.random.seed(s)`` before
                       ``run`` makes a simulation repeatable, as it does for the reference).
    keep_trajectories  'auto' | True | False: materialise sensors + outputs in HBM ('auto': when they fit
                       ``max_device_bytes``), else keep only per-run end-point errors (stats-only).
    max_device_bytes   budget for materialised series on one GPU (default 64 GiB of the 288 GB).
    keep_runs          stats-only mode: still materialise sensors (incl. GPS / magnetometer), outputs and CSV files of the
                       first K runs of this rank (the counter RNG makes them the same runs wherever they are integrated: as the
                       first 256-run workgroup of the batch on a sibling stream when K <= 256 and this process integrates on one
                       GPU in fp64 -- no time of their own --, otherwise in a second, small launch).
    stats_start        stats-only mode: the ``err_stats_start`` that ``results()`` will be asked for (default 0 s, the
                       reference's default; -1 = end-point only).  The process-error statistics of that window are
                       accumulated inside the kernel; asking ``results()`` for another window integrates once more.
    device             GPU index (default LOCAL_RANK or 0).
    devices            'all' | [ids]: spread the runs of THIS process over several GPUs -- one context and one Python thread per
                       entry, contiguous run ranges, per-device records folded with the library's Chan merge (ginsim.multi;
                       no torch, no launcher).  The reference's loop being sharded is ins_sim.py:490-506.  An id may repeat
                       (``devices=[0, 0]``: two contexts on one GPU).  Not together with torch.distributed, where the
                       split is one process per GPU.  Default (None): $GINSIM_DEVICES if set (same values, comma separated;
                       'one' = never spread), else EVERY visible GPU when the batch is large enough to pay for it
                       (sim_count x samples >= 2^30, e.g. BASELINE configs 3 and 4) and nothing says this process owns one
                       GPU only (no device=, no $LOCAL_RANK, no initialised process group) -- so that an UNCHANGED
                       demo_free_integration.py uses the whole node.
    placed             None (default) | True | False: carve the materialised series from the device's PLACED arena (ginsim.Context
                       placed_*, ABI 7: a range whose 512 MiB stripes cycle through the three 96 GB classes of an MI355X's memory,
                       built once per device and process from physical chunks whose class was measured) -- C2's launch then takes
                       1.15-1.2 ms wherever the process stands instead of 1.21-1.40 ms by where hipMalloc put the planes.  Default:
                       on for batches that materialise 1 GiB or more; $GINSIM_PLACED=0 switches it off.  ``sim.placement`` says
                       what the last run() got.  (``spread_outputs=`` of round 5 is accepted as an alias.)
    geo_mag_n          geomagnetic field [uT] in the N frame at the initial position, needed for a 9-axis IMU.  The
                       reference evaluates the WMM model once per run for this vector (pathgen.py:164-168,
                       date = today); that model is outside the accelerated path: either the caller supplies the vector, or
                       a checkout of the reference is named in $GNSS_INS_SIM_REFERENCE and ITS geomag.py is
                       evaluated once on the host (geoparams.reference_geomag_n; geo_mag_date pins the date).
"""
import math
import os
import sys
import time

import numpy as np

from .ins_data_manager import InsDataMgr
from .ins_algo_manager import InsAlgoMgr
from . import sim_data
from .sim_data import McSeries, ChainSeries
from ..attitude import attitude

NAME = 'gnss-ins-sim'
VERSION = '3.0.0_alpha'
high_mobility = np.array([1.0, 0.5, 2.0])       # m/s/s, rad/s/s, rad/s  (ins_sim.py:25)


KEPT_BLOCK = 256        # runs of one workgroup of the lane-per-run kernels


class _BlockAndRest(object):
    """The statistics job of a statistics-only Sim with kept runs, in two launches that run AT THE SAME TIME: `block` integrates the
    first KEPT_BLOCK runs with everything materialised (the kept runs are its first ones) on a sibling context, `rest` every other
    run, statistics only -- together as many workgroups as the one launch over all runs, so the kept runs cost no time of their own
    (C3: the 2-run launch was a chain of 193 036 dependent steps, 0.27 s next to 0.99 s; tools/experiments/kept_block_overlap.py:
    block || rest 0.990 s against 0.991 s for the one launch).  Answers what _McResults asks of a statistics job; the per-run
    records are the block's followed by the rest's (run order), the end-point records are folded with the library's Chan merge."""

    keep_traj = False       # as a statistics job: the trajectories of the block are reached through Sim's kept jobs

    def __init__(self, block, rest):
        self.block, self.rest = block, rest
        self.precision, self.n, self.algos = rest.precision, rest.n, rest.algos
        self.runs = block.runs + rest.runs
        self.proc_first, self.proc_ned, self.end_ned = rest.proc_first, rest.proc_ned, rest.end_ned

    def stats(self, algo, ned=False):
        import ginsim
        return ginsim.StatsResult.merge([self.block.stats(algo, ned=ned).pack(), self.rest.stats(algo, ned=ned).pack()])

    def end_errors(self, algo, ned=False):
        return np.concatenate([self.block.end_errors(algo, ned=ned), self.rest.end_errors(algo, ned=ned)], axis=0)

    def process_stats_online(self, algo):
        return np.concatenate([self.block.process_stats_online(algo), self.rest.process_stats_online(algo)], axis=0)

    def release(self):
        self.rest.release()     # the block stays: it is also the kept job


class _McResults(object):
    """Device results of one Sim.run: per-algorithm jobs + cross-rank merge of the statistics.

    jobs[i] integrates ALL runs of this rank for fused algorithm i (statistics); kept[i] is the job whose series are
    materialised (the same object when everything is kept, a job over the first ``keep_runs`` runs otherwise, or None).
    """

    def __init__(self, jobs, kept, names, kinds, first_run, runs_local, total_runs, group, device, make_ps_job=None, ctx=None,
                 make_kept_job=None, block_runs=0, ned_from_traj=False):
        self.jobs, self.kept, self.algo_names, self.kinds = jobs, kept, names, kinds
        # does the NED end-point record have to be recomputed from trajectories?  Decided by Sim from the CONFIGURATION
        # (identical on every rank), never from a rank's own jobs: a rank without runs has none, and the ranks must enter the
        # same collective
        self._ned_needs_traj = bool(ned_from_traj)
        self.first_run, self.runs_local, self.total_runs = first_run, runs_local, total_runs
        self._group, self._device, self._stats = group, device, {}
        self._make_ps_job, self._ctx = make_ps_job, ctx
        self._make_kept_job, self._block_runs = make_kept_job, int(block_runs)
        self.exchange = None                # which exchange merged the records: 'abi', 'torch (...)', None = one process

    def job_of(self, name):
        return self.jobs[self.algo_names.index(name)]

    def end_stats(self, name, ned=False):
        key = (name, bool(ned))
        if key not in self._stats:
            import ginsim
            from ginsim import distributed
            job, kind = self.job_of(name), self.kinds[self.algo_names.index(name)]
            # every rank takes the same branch: the flag comes from the Sim configuration, which is the same on every rank; a
            # rank without runs (world > sim_count) has job None and contributes the empty record to the SAME collective
            from_traj = ned and self._ned_needs_traj
            if from_traj:                           # NED record recomputed from trajectories (kept, or re-integrated block by block)
                part = ginsim.StatsResult.zero() if job is None else self._ned_from_traj(self.algo_names.index(name))
                self._stats[key] = part if self._group is None else distributed.allreduce_stats(part, self._group, self._device)
            elif self._group is None:
                self._stats[key] = job.stats(kind, ned=ned)
            else:
                # the record is on the device: the library's own RCCL all-gather behind the C ABI when the backend is nccl
                # (the exchange bench.py times), the torch.distributed all-reduce otherwise
                ex = distributed.StatsExchange.of(self._ctx, self._group, self._device)
                self._stats[key] = ex.merge(job, kind, ned=ned)
                self.exchange = ex.kind if ex.note is None else '%s (%s)' % (ex.kind, ex.note)
        return self._stats[key]

    def _blocks(self, idx):
        """Statistics that need trajectories a stats-only launch did not keep (the fp32 kernel has no online accumulator): the
        runs are integrated again in blocks that fit the device budget, trajectories kept, and each block is reduced on the
        device before the next one is launched -- the counter RNG reproduces exactly the same runs."""
        step = max(self._block_runs, 1)
        for off in range(0, self.runs_local, step):
            blk = self._make_kept_job(idx, off, min(step, self.runs_local - off))
            blk.run()
            try:
                yield blk
            finally:
                blk.release()

    def _ned_from_traj(self, idx):
        import ginsim
        job, kind = self.jobs[idx], self.kinds[idx]
        if job.keep_traj:
            return job.stats_from_traj(kind, pos_ned=True)
        return ginsim.StatsResult.merge([b.stats_from_traj(kind, pos_ned=True).pack() for b in self._blocks(idx)])

    def _process_array(self, idx, start_sample, ned):
        """(runs, 3, 9) process statistics of fused algorithm idx over this rank's runs."""
        job, kind = self.jobs[idx], self.kinds[idx]
        key = ('proc', idx, int(start_sample), bool(ned))
        if key not in self._stats:
            if job.precision != 'f64' and not job.keep_traj:       # fp32, statistics only: re-integrate block by block
                self._stats[key] = np.concatenate([b.process_stats(kind, start_sample, pos_ned=ned) for b in self._blocks(idx)])
            elif job.keep_traj:
                self._stats[key] = job.process_stats(kind, start_sample, pos_ned=ned)
            elif job.proc_first == int(start_sample) and job.proc_ned == bool(ned):
                self._stats[key] = job.process_stats_online(kind)
            else:       # another window than the one run() accumulated: integrate again with that window (same counter
                        # RNG, same runs, nothing kept) -- costs one more launch
                ps = self._make_ps_job(idx, int(start_sample), bool(ned))
                ps.run()
                self._stats[key] = ps.process_stats_online(kind)
                ps.release()
        return self._stats[key]

    def process_stats(self, data_name, start_sample, ned=False):
        """{'max'|'avg'|'std': {'<algo>_<run>': (3,)}} for data_name in att_euler/pos/vel (ins_data_manager.py:761-795)."""
        from .sim_data import RunStats
        sl = {'att_euler': slice(0, 3), 'pos': slice(3, 6), 'vel': slice(6, 9)}[data_name]
        parts = {'max': [], 'avg': [], 'std': []}
        for idx, name in enumerate(self.algo_names):
            if self.jobs[idx] is None:
                continue
            arr = self._process_array(idx, start_sample, ned)
            for row, s in enumerate(('max', 'avg', 'std')):
                parts[s].append((name, self.first_run, arr[:, row, sl]))
        return {s: RunStats(parts[s]) for s in parts}

    def run_of_key(self, key):
        return int(str(key).rsplit('_', 1)[-1]) if isinstance(key, str) else int(key)


class Sim(object):
    def __init__(self, fs, motion_def, ref_frame=0, imu=None, mode=None, env=None, algorithm=None, *,
                 seed=None, keep_trajectories='auto', max_device_bytes=64 * 2 ** 30, device=None, geo_mag_n=None, precision='f64',
                 keep_runs=0, stats_start=0, geo_mag_date=None, devices=None, placed=None, spread_outputs=None):
        self.name, self.version = NAME, VERSION
        self.fs, self.imu, self.mode, self.env = fs, imu, mode, env
        self.ref_frame = ref_frame if ref_frame in (0, 1) else 0
        self.sim_count = 1
        self.sim_complete = False
        self.sim_results = False
        self.dmgr = InsDataMgr(fs, self.ref_frame)
        self.data_src = motion_def
        self.data_from_files = False
        self.amgr = InsAlgoMgr(algorithm)
        self.interested_error = {'att_euler': 'angle', 'pos': None, 'vel': None}
        self.sum = ''
        self.seed, self.keep_trajectories, self.max_device_bytes, self.device = seed, keep_trajectories, max_device_bytes, device
        self.geo_mag_n, self.geo_mag_date = geo_mag_n, geo_mag_date
        self.precision = precision      # 'f32': single-precision kernel (tolerances: tests/test_gpu_fp32.py)
        self.keep_runs, self.stats_start = max(int(keep_runs), 0), stats_start
        self.placed = placed if placed is not None else (None if spread_outputs is None else bool(spread_outputs))
        self.placement = None            # MonteCarloJob.placement() of the last materialising run
        self._auto_devices = False
        if devices is None and device is None:
            env = os.environ.get('GINSIM_DEVICES', '').strip()
            if env and env != 'one':
                devices = env if env == 'all' else [int(x) for x in env.split(',') if x.strip()]
            elif not env:
                self._auto_devices = True           # decided per run(), when the size of the batch is known
        self.devices = devices
        self._devset = None
        self._side_ctx = None
        self.mc = None

    # ------------------------------------------------------------------------------------ run
    def run(self, num_times=1):
        self.sim_count = max(int(num_times), 1)
        if isinstance(self.data_src, str) and os.path.isdir(self.data_src):
            self._run_from_files()
        else:
            self._psd_on_grid = []
            try:
                self._run_monte_carlo()
            finally:
                for v in self._psd_on_grid:
                    for k in 'xyz':
                        v[k][1:-1] *= 0.5 ** self.sim_count
        self.sim_complete = True

    AUTO_SPREAD_WORK = 2 ** 30      # sample x run products from which an un-configured Sim uses every visible GPU

    def _context(self, work=0, distributed=False):
        """Where this process integrates: a ginsim.Context (one GPU) or a ginsim.multi.DeviceSet (devices=..., or by default
        every visible GPU for a batch of at least AUTO_SPREAD_WORK sample x run products when this process is not one rank
        of a one-process-per-GPU job)."""
        import ginsim
        if self.devices is None and self._auto_devices and not distributed and not self._launcher_rank() \
                and work >= self.AUTO_SPREAD_WORK and ginsim.device_count() > 1:
            self.devices = 'all'
            if not Sim._AUTO_SPREAD_SAID:       # once per process: an unchanged script should not take a node silently
                Sim._AUTO_SPREAD_SAID = True
                print('gnss_ins_sim: %d sample x run products -> spreading the runs over all %d visible GPUs (GINSIM_DEVICES=one '
                      'keeps one GPU, GINSIM_DEVICES=0,1 names them; statistics merged across devices equal a one-GPU run to '
                      'rounding, not to the bit)' % (work, ginsim.device_count()), file=sys.stderr)
        if self.devices is not None:
            from ginsim import multi
            if self.device is not None:
                raise ValueError('Sim: give device (one GPU) or devices (several), not both')
            if self._devset is None or self._devset.devices != multi.parse_devices(self.devices):
                self._devset = multi.DeviceSet(self.devices)
            return self._devset
        if self.device is None:
            return ginsim.default_context()
        return ginsim.Context(self.device)

    _AUTO_SPREAD_SAID = False
    _RANK_ENV = ('LOCAL_RANK', 'OMPI_COMM_WORLD_LOCAL_RANK', 'SLURM_LOCALID', 'PMI_RANK', 'PMIX_RANK', 'MV2_COMM_WORLD_LOCAL_RANK')

    @staticmethod
    def _launcher_rank():
        """True when the environment says this process is ONE RANK of a multi-process job (torchrun, mpirun, srun ...): such a
        process owns one GPU and must not spread over the node by itself."""
        return any(k in os.environ for k in Sim._RANK_ENV)

    _SIBLINGS = {}          # device -> a second context (its own stream) for launches that run next to the main one

    def _block_and_rest_contexts(self, ctx, rest_workgroups):
        """(context of the one-workgroup launch of the kept runs, context of the statistics launch over the others): `ctx` and a
        sibling context of the same device, which way round decided by where their launches START.
        The dispatcher deals the workgroups of a launch to the eight XCDs of an MI355X in turn, starting at a die that belongs to the
        stream's hardware queue (ginsim_stream_first_xcc; consecutive for streams made one after the other, shifted by every stream
        any library of the process made in between -- the FFT plans of a PSD vibration, say).  A statistics launch of W workgroups on
        8 x 64 slots leaves a slot to spare on die (first + W) mod 8 when W is not a multiple of 8; the kept runs' workgroup on THAT
        die costs nothing (C3: 0.93 s for the pair), on any other die it costs its die a third round of workgroups (1.20 s;
        tools/experiments/c3_pair_matrix.py: 24 of 24 pairs as this rule says)."""
        import ginsim
        want = int(rest_workgroups) % 8
        hit = Sim._SIBLINGS.get(ctx.device)
        if hit is not None and hit['of'] == ctx.handle and hit['want'] == want and hit['side'].handle is not None:
            self._side_ctx = hit['side']
            return hit['pair']
        # Candidates first, questions afterwards: the dies move while the runtime is still making its hardware queues (four by
        # default: a fresh process answers 0 for every stream, then 0 / 7 / 6 / 5 ... as the queues come; with all of them there
        # the answers stay).  The candidates that are not taken stay alive -- closing streams could give queues back.
        spare = hit['spare'] + [hit['side']] if hit is not None and hit['of'] == ctx.handle else []
        spare = [c for c in spare if c.handle is not None and c is not ctx]
        try:                                # one more stream than the runtime has hardware queues: all of them exist afterwards
            want_spare = 1 + max(1, int(os.environ.get('GPU_MAX_HW_QUEUES', '4')))
        except ValueError:
            want_spare = 5
        while len(spare) < want_spare:
            spare.append(ginsim.Context(ctx.device))
        for c in spare:
            c.first_xcc()
        mine = ctx.first_xcc()
        pair = side = None
        for c in spare:
            theirs = c.first_xcc()
            if theirs == (mine + want) % 8:
                pair = (c, ctx)             # the kept runs on the sibling
            elif mine == (theirs + want) % 8:
                pair = (ctx, c)             # the kept runs here, the statistics launch on the sibling
            if pair is not None:
                side = c
                break
        if pair is None:                    # no such pair among the queues: a sibling on ANOTHER die at least (the same die = the same
            other = [c for c in spare if c.first_xcc() != mine]         # hardware queue: the two launches one after the other);
            side = (other or spare)[0]                                  # it costs one die one more round of workgroups
            pair = (side, ctx)
        spare = [c for c in spare if c is not side]
        Sim._SIBLINGS[ctx.device] = {'of': ctx.handle, 'want': want, 'side': side, 'pair': pair, 'spare': spare}
        self._side_ctx = side
        return pair

    @staticmethod
    def _dist():
        """(rank, world, group, exchange device) of the torch.distributed job, or a single-process stand-in."""
        # a process group can only exist if the caller imported torch.distributed already: do not pay the ~1 s
        # `import torch` of a single-GPU script for a question whose answer is then known to be "no"
        dist = sys.modules.get('torch.distributed')
        if dist is None or not (dist.is_available() and dist.is_initialized()):
            return 0, 1, None, None
        import torch
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if dist.get_backend() == 'nccl' \
            else torch.device('cpu')
        return dist.get_rank(), dist.get_world_size(), dist.group.WORLD, dev

    def _pick_seed(self, group, dev):
        seed = self.seed
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 62))
        if group is not None:                      # every rank must use rank 0's key
            import torch
            import torch.distributed as dist
            t = torch.tensor([seed], dtype=torch.int64, device=dev)
            dist.broadcast(t, src=0, group=group)
            seed = int(t.item())
        return seed

    def _run_monte_carlo(self):
        import ginsim
        from ginsim import workloads, distributed
        if self.imu is None:
            raise ValueError('an IMU model is required to generate sensor data from a motion definition')
        ini_pva, motion_def = workloads.parse_motion(self.data_src)
        mobility = self._parse_mode(self.mode)
        fs_imu = self.fs[0]
        if self.imu.magnetometer and self.geo_mag_n is None:
            # pathgen.py:164-168 evaluates the World Magnetic Model ONCE, on the host, at the initial position: done here with the
            # reference's own geomag.py when a reference checkout is reachable (unchanged 9-axis scripts then run unchanged)
            from ..geoparams import geoparams
            self.geo_mag_n = geoparams.reference_geomag_n(ini_pva[0], ini_pva[1], ini_pva[2], self.geo_mag_date)
            if self.geo_mag_n is None:
                raise NotImplementedError('a 9-axis IMU needs the local geomagnetic field: pass Sim(..., geo_mag_n=[bx,by,bz] uT), or '
                                          'name a checkout of the reference (its gnss_ins_sim/geoparams/geomag.py + WMM.COF) '
                                          'in $GNSS_INS_SIM_REFERENCE (the WMM evaluation of pathgen.py:164-168 is outside the '
                                          'accelerated path)')
        raw = ginsim.pathgen(ini_pva, motion_def, fs_imu, self.fs[1] if self.imu.gps else 0.0, mobility,
                             self.ref_frame, gps=self.imu.gps,
                             geo_mag_n=self.geo_mag_n if self.imu.magnetometer else None)
        d = self.dmgr
        nav, imu_t = raw['nav'], raw['imu']
        n = nav.shape[0]
        d.add_data(d.time.name, nav[:, 0] / fs_imu)                       # ins_sim.py:467-480
        d.add_data(d.ref_pos.name, np.ascontiguousarray(nav[:, 1:4]))
        d.add_data(d.ref_vel.name, np.ascontiguousarray(nav[:, 4:7]))
        d.add_data(d.ref_att_euler.name, np.ascontiguousarray(nav[:, 7:10]))
        d.add_data(d.ref_accel.name, np.ascontiguousarray(imu_t[:, 1:4]))
        d.add_data(d.ref_gyro.name, np.ascontiguousarray(imu_t[:, 4:7]))
        if self.imu.gps:
            d.add_data(d.gps_time.name, raw['gps'][:, 0] / fs_imu)
            d.add_data(d.ref_gps.name, np.ascontiguousarray(raw['gps'][:, 1:7]))
            d.add_data(d.gps_visibility.name, raw['gps'][:, 7].copy())
        if self.imu.magnetometer:
            d.add_data(d.ref_mag.name, np.ascontiguousarray(raw['mag'][:, 1:4]))
        if self.imu.odo:
            d.add_data(d.ref_odo.name, np.ascontiguousarray(raw['odo'][:, 2]))
        d.add_data(d.ref_att_quat.name, sim_data.Lazy(lambda e=d.ref_att_euler.data.copy(): attitude.euler2quat(e)))    # ins_sim.py:729-748, on first read
        truth = {'ref_accel': d.ref_accel.data, 'ref_gyro': d.ref_gyro.data, 'ref_pos': d.ref_pos.data,
                 'ref_vel': d.ref_vel.data, 'ref_att': d.ref_att_euler.data}
        if self.imu.odo:
            truth['ref_odo'] = d.ref_odo.data

        # which plugins are inside the fused kernel
        algos = self.amgr.algo or []
        kinds = [getattr(a, 'mc_algo', None) for a in algos]
        fused = [i for i, k in enumerate(kinds) if k in ('free', 'odo')]
        hosted = [i for i in range(len(algos)) if i not in fused]
        for i in fused:
            if kinds[i] == 'odo' and not self.imu.odo:
                raise ValueError("algorithm %d needs 'odo' but the IMU model has no odometer" % i)

        # environment --> vibration parameters (ins_sim.py:482-489): 'random' and 'sinusoidal' models are terms of the kernels, a
        # 'psd' model is a series per run and axis made on the device before the launch (ginsim_vib_psd_series)
        vib_acc = vib_gyro = None
        if self.env is not None:
            if 'acc' in self.env.keys():
                vib_acc = self._parse_env(self.env['acc'])
            if 'gyro' in self.env.keys():
                vib_gyro = self._parse_env(self.env['gyro'])
            for v in (vib_acc, vib_gyro):
                if v is not None and v['type'] != 'psd':
                    ginsim.vibration(v, fs_imu, False)       # raises for a definition the kernels do not know
                elif v is not None and self.precision != 'f64':
                    raise NotImplementedError("the 'psd' vibration (an (n, 4) env array) runs on the fp64 kernels only")
        vib = dict(vib_accel=vib_acc, vib_gyro=vib_gyro)
        # A PSD given on the series' own frequency grid is halved IN PLACE by the reference at every run and axis
        # (time_series_from_psd.py:44-49: no copy is made when no interpolation is needed; the arrays are views of the caller's
        # env).  The device applies run r's factor 0.5^(r + 1); the arrays are left as the reference leaves them, after the run.
        self._psd_on_grid = [v for v in (vib_acc, vib_gyro) if v is not None and v['type'] == 'psd' and
                             (ginsim.psd_amplitudes(v, fs_imu, n) or (0, 0, False))[2]]

        rank, world, group, xdev = self._dist()
        first, count = distributed.shard(self.sim_count, world, rank)
        seed = self._pick_seed(group, xdev)
        ctx = self._context(self.sim_count * n, group is not None)     # one GPU (Context) or several (multi.DeviceSet)
        from ginsim import multi
        spread = isinstance(ctx, multi.DeviceSet)
        ndev = len(ctx) if spread else 1
        if spread and group is not None:
            raise ValueError('Sim(devices=...) spreads the runs of ONE process over several GPUs; under torch.distributed the '
                             'split is one process per GPU (drop devices=, or do not initialise a process group)')
        new_job = (lambda *a, **kw: multi.JobSet(ctx, *a, **kw)) if spread else (lambda *a, **kw: ginsim.MonteCarloJob(ctx, *a, **kw))
        per_sample = 48 + (8 if self.imu.odo else 0) + 72 * len(fused) + (24 if self.imu.magnetometer else 0) + \
            (48.0 * raw['gps'].shape[0] / n if self.imu.gps else 0)
        keep = self.keep_trajectories
        if keep == 'auto':
            # decided on the LARGEST share of any rank / device (rank 0's), so that every rank takes the same decision -- the
            # ranks enter collectives that depend on it
            largest = -(-distributed.shard(self.sim_count, world, 0)[1] // ndev)
            keep = per_sample * n * max(largest, 1) <= self.max_device_bytes
        if hosted and not keep:
            raise ValueError('plugins outside the fused kernel need the sensor series: use keep_trajectories=True')
        self.kept = bool(keep)

        # one launch per distinct (initial states, earth_rot); identical noise in every launch (counter RNG)
        groups = []
        for i in fused:
            a = algos[i]
            for g in groups:
                if np.array_equal(g['ini'], a.ini) and g['earth_rot'] == a.earth_rot and kinds[i] not in g['kinds'] \
                        and g['first'] == a.run_times:
                    g['kinds'].append(kinds[i])
                    g['idx'].append(i)
                    break
            else:
                groups.append({'ini': a.ini, 'earth_rot': a.earth_rot, 'kinds': [kinds[i]], 'idx': [i], 'first': a.run_times})
        # Which runs have their series materialised: all of this rank's runs (keep), or the first keep_runs of them next to
        # a stats-only launch over all runs (the counter RNG makes the small launch reproduce exactly those runs).
        kcount = count if keep else min(self.keep_runs, count)
        t_axis = nav[:, 0] / fs_imu

        def sample_of(start_s):
            hit = np.where(t_axis >= max(float(start_s), 0.0))[0]
            return int(hit[0]) if hit.shape[0] else 0

        def make_job(g, kinds_, runs_, keep_sens, keep_traj, **kw):
            return new_job(fs_imu, self.ref_frame, truth, self.imu.accel_err, self.imu.gyro_err,
                           g['ini'], runs=runs_, algos=tuple(kinds_), odo_err=self.imu.odo_err,
                           earth_rot=g['earth_rot'], seed=seed, run_offset=first,
                           ini_first=g['first'] + first, keep_sensors=keep_sens, keep_traj=keep_traj,
                           precision=self.precision, placed=self.placed, **vib, **kw)

        f64 = self.precision == 'f64'
        online = (not keep) and f64 and self.stats_start is not None and self.stats_start != -1
        end_ned = (not keep) and f64 and self.ref_frame == 0
        stats_jobs, kept_jobs, group_of = {}, {}, {}
        sensor_job = None
        if count > 0:
            for g in groups:
                for i in g['idx']:
                    group_of[i] = g
                if keep:
                    job = make_job(g, g['kinds'], count, sensor_job is None, True)
                    job.launch()
                    sensor_job = sensor_job or job
                    for i in g['idx']:
                        stats_jobs[i] = kept_jobs[i] = job
                    continue
                # The kept runs as the FIRST WORKGROUP of the batch (one process, one GPU, fp64): a block of KEPT_BLOCK runs with
                # everything materialised on a sibling context, at the same time as the statistics-only launch over the other runs
                # (_BlockAndRest).  Otherwise: one small launch for them in front of the launch over all runs.
                ride = (0 < kcount <= KEPT_BLOCK < count and f64 and group is None and not spread and
                        per_sample * n * KEPT_BLOCK <= self.max_device_bytes)
                if ride:
                    # which of the two contexts takes which launch: by where their launches start (_block_and_rest_contexts)
                    c_block, c_rest = self._block_and_rest_contexts(ctx, -(-(count - KEPT_BLOCK) // KEPT_BLOCK))
                    per_launch = [[k] for k in g['kinds']] if online else [list(g['kinds'])]
                    for kinds_ in per_launch:
                        kw = dict(proc_first=sample_of(self.stats_start)) if online else {}
                        blk = ginsim.MonteCarloJob(c_block, fs_imu, self.ref_frame, truth, self.imu.accel_err, self.imu.gyro_err, g['ini'],
                                                   runs=KEPT_BLOCK, algos=tuple(kinds_), odo_err=self.imu.odo_err, earth_rot=g['earth_rot'],
                                                   seed=seed, run_offset=first, ini_first=g['first'] + first,
                                                   keep_sensors=sensor_job is None, keep_traj=True, precision=self.precision,
                                                   end_ned=end_ned, **vib, **kw)
                        rest = ginsim.MonteCarloJob(c_rest, fs_imu, self.ref_frame, truth, self.imu.accel_err, self.imu.gyro_err, g['ini'],
                                                    runs=count - KEPT_BLOCK, algos=tuple(kinds_), odo_err=self.imu.odo_err,
                                                    earth_rot=g['earth_rot'], seed=seed, run_offset=first + KEPT_BLOCK,
                                                    ini_first=g['first'] + first + KEPT_BLOCK, precision=self.precision,
                                                    end_ned=end_ned, **vib, **kw)
                        blk.launch()
                        rest.launch()
                        sensor_job = sensor_job or blk
                        both = _BlockAndRest(blk, rest)
                        for i, kind in zip(g['idx'], g['kinds']):
                            if kind in kinds_:
                                stats_jobs[i], kept_jobs[i] = both, blk
                    continue
                if kcount > 0:  # the kept runs: one small launch
                    kj = make_job(g, g['kinds'], kcount, sensor_job is None, True)
                    kj.launch()
                    sensor_job = sensor_job or kj
                    for i in g['idx']:
                        kept_jobs[i] = kj
                if online:      # process-error statistics accumulated inside the kernel: one algorithm per launch
                    for i, kind in zip(g['idx'], g['kinds']):
                        stats_jobs[i] = make_job(g, [kind], count, False, False, proc_first=sample_of(self.stats_start),
                                                 end_ned=end_ned)
                        stats_jobs[i].launch()
                else:
                    job = make_job(g, g['kinds'], count, False, False, end_ned=end_ned)
                    job.launch()
                    for i in g['idx']:
                        stats_jobs[i] = job
            if not groups and kcount > 0:      # Sim without algorithm: sensor generation only (demo_no_algo.py)
                sensor_job = new_job(fs_imu, self.ref_frame, truth, self.imu.accel_err,
                                     self.imu.gyro_err, None, runs=kcount, algos=(), odo_err=self.imu.odo_err,
                                     seed=seed, run_offset=first, keep_sensors=True, **vib)
                sensor_job.launch()
            ctx.sync()
            if self._side_ctx is not None:
                self._side_ctx.sync()
            self.placement = sensor_job.placement() if sensor_job is not None and hasattr(sensor_job, 'placement') else None
        for i in fused:                         # FreeIntegration.run_times accounting (free_integration.py:69)
            algos[i].run_times += self.sim_count

        # expose device series through the data manager
        runs = range(first, first + kcount)
        in_kept = lambda k: int(k) - first if isinstance(k, (int, np.integer)) and first <= int(k) < first + kcount else None
        if sensor_job is not None:
            def sens(name, squeeze=False, job=sensor_job):
                return McSeries(kcount, lambda pos, j=job, nm=name: j.sensors(nm, pos)[..., None] if nm == 'odo'
                                else j.sensors(nm, pos), key_of=lambda i: first + i, pos_of=in_kept, squeeze=squeeze)
            d.add_data(d.accel.name, sens('accel'))
            d.add_data(d.gyro.name, sens('gyro'))
            if self.imu.odo:
                d.add_data(d.odo.name, sens('odo', squeeze=True))
        if kcount > 0 and (self.imu.gps or self.imu.magnetometer):      # ins_sim.py:497-503
            aux = (multi.AuxJobSet if spread else ginsim.AuxSensorJob)(
                ctx, kcount, seed=seed, run_offset=first,
                ref_gps=d.ref_gps.data if self.imu.gps else None, gps_err=self.imu.gps_err, ref_frame=self.ref_frame,
                ref_mag=d.ref_mag.data if self.imu.magnetometer else None, mag_err=self.imu.mag_err).run()
            self._aux = aux
            view = lambda nm: McSeries(kcount, lambda pos, a=aux, nm=nm: a.series(nm, pos), key_of=lambda i: first + i,
                                       pos_of=in_kept)
            if self.imu.gps:
                d.add_data(d.gps.name, view('gps'))
            if self.imu.magnetometer:
                d.add_data(d.mag.name, view('mag'))
        if self.amgr.algo is not None:
            d.set_algo_output(self.amgr.output)
        names = [self.amgr.get_algo_name(i) for i in fused]
        if fused and kcount > 0:
            for out_name, comp in (('att_euler', 0), ('pos', 1), ('vel', 2)):
                d.add_data(out_name, self._output_view(kept_jobs, fused, kinds, names, comp, first, kcount))
            d.add_data('att_quat', self._output_view(kept_jobs, fused, kinds, names, 0, first, kcount, quat=True))
        elif fused:
            for out_name in ('att_euler', 'pos', 'vel'):       # stats-only: names are known, series are not kept
                d.add_data(out_name, {})
        if fused:
            def make_ps_job(idx, start_sample, ned):
                i = fused[idx]
                return make_job(group_of[i], [kinds[i]], count, False, False, proc_first=start_sample,
                                proc_ned=ned, end_ned=False)
            def make_kept_job(idx, off, runs_):      # a block of this rank's runs, trajectories kept (fp32 statistics)
                i = fused[idx]
                g = group_of[i]
                return new_job(fs_imu, self.ref_frame, truth, self.imu.accel_err, self.imu.gyro_err, g['ini'],
                               runs=runs_, algos=(kinds[i],), odo_err=self.imu.odo_err, earth_rot=g['earth_rot'],
                               seed=seed, run_offset=first + off, ini_first=g['first'] + first + off,
                               keep_sensors=False, keep_traj=True, precision=self.precision, **vib)
            esize = 4 if self.precision == 'f32' else 8
            block_runs = ndev * max(256, int(self.max_device_bytes // (9 * esize * n)) // 256 * 256)
            self.mc = _McResults([stats_jobs.get(i) for i in fused], [kept_jobs.get(i) for i in fused], names,
                                 [kinds[i] for i in fused], first, count, self.sim_count, group, xdev, make_ps_job, ctx=ctx,
                                 make_kept_job=make_kept_job, block_runs=block_runs, ned_from_traj=not end_ned)
            self.mc.devices = list(ctx.devices) if spread else None
            self.mc.kept_block = any(isinstance(j, _BlockAndRest) for j in stats_jobs.values())    # the kept runs rode along
            d.set_mc_results(self.mc)
        # plugins outside the fused kernel: the reference's per-run loop over host copies (user code)
        if hosted:
            # plugins that take the device-resident sensor series of all runs at once (demo_algorithms.allan_analysis)
            on_device = [i for i in hosted if hasattr(algos[i], 'run_device') and sensor_job is not None and count > 0]
            merged = [{} for _ in self.amgr.output]
            for i in on_device:
                name = self.amgr.get_algo_name(i)
                per_run = algos[i].run_device(sensor_job, fs_imu)
                for k, res in enumerate(per_run):
                    for j, slot in enumerate(self.amgr.output_alloc[i]):
                        merged[slot][name + '_' + str(first + k)] = res[j]
            rest = [i for i in hosted if i not in on_device]
            if rest:
                inputs = d.get_data(self.amgr.input)
                out = self.amgr.run_algo(inputs, list(runs), only=rest)
                for j in range(len(merged)):
                    merged[j].update(out[j])
            for j, oname in enumerate(self.amgr.output):
                if merged[j]:
                    cur = d.get_data_all(oname).data if oname in d.available else None
                    if isinstance(cur, McSeries):           # a fused plugin produced this output too: keep its device view
                        d.add_data(oname, ChainSeries(cur, merged[j]))
                    else:
                        d.add_data(oname, merged[j])

    def _output_view(self, jobs_by_algo, fused, kinds, names, comp, first, count, quat=False):
        """Mapping '<algo>_<run>' -> (n,3) (or (n,4) quaternion) over the trajectory buffers of all fused plugins."""
        order = [(names[k], jobs_by_algo[i], kinds[i]) for k, i in enumerate(fused)]

        def locate(key):
            if not isinstance(key, str) or '_' not in key:
                return None
            nm, _, r = key.rpartition('_')
            if not r.isdigit() or not (first <= int(r) < first + count):
                return None
            for a, (name, _, _) in enumerate(order):
                if name == nm:
                    return a * count + int(r) - first
            return None

        def fetch(positions):
            out = []
            for p in positions:
                a, r = divmod(p, count)
                x = order[a][1].trajectories(order[a][2], [r])[comp][0]
                out.append(attitude.euler2quat(x) if quat else x)
            return np.stack(out)

        return McSeries(count * len(order), fetch, key_of=lambda p: order[p // count][0] + '_' + str(first + p % count),
                        pos_of=locate)

    # ------------------------------------------------------------------------------------ logged data
    def _run_from_files(self):
        """Sim with a directory of logged CSV files (ins_sim.py:426-442): every plugin runs once per data key on
        the given-data entry point of the library."""
        self.data_src = os.path.abspath(self.data_src)
        self.data_from_files = True
        d = self.dmgr
        for fname in sorted(os.listdir(self.data_src)):
            low = fname.lower()
            if not low.endswith('.csv'):
                continue
            name, key = low[:-4], None
            cut = name.rfind('-')
            if cut != -1:
                key = name[cut + 1:]
                name = name[:cut]
                key = int(key) if key.isdigit() else key
            if not d.is_supported(name):
                continue
            full = os.path.join(self.data_src, fname)
            data = np.genfromtxt(full, delimiter=',', skip_header=1)
            with open(full) as f:
                cols = f.readline().split(',')
            units = [c[c.find('(') + 1:c.rfind(')')] for c in cols if '(' in c and c.rfind(')') > c.find('(')]
            units = units if len(units) == len(cols) else None
            if name in ('ref_pos', 'pos') and self.ref_frame == 1 and units in (['deg', 'deg', 'm'], ['rad', 'rad', 'm']):
                raise NotImplementedError('LLA -> local-frame conversion of logged positions is outside the hot path')
            d.add_data(name, data, key, units)
        if self.amgr.algo is not None:
            d.set_algo_output(self.amgr.output)
            inputs = d.get_data(self.amgr.input)
            out = self.amgr.run_algo(inputs, range(self.sim_count))
            for j, oname in enumerate(self.amgr.output):
                d.add_data(oname, out[j])

    # ------------------------------------------------------------------------------------ results
    def results(self, data_dir=None, err_stats_start=0, gen_kml=False, extra_opt='', *, max_saved_runs=16, max_summary_runs=None):
        """Sim.results (ins_sim.py:194-251).  CSV files are written for at most ``max_saved_runs`` Monte-Carlo runs.  The printed
        summary lists the per-run process statistics of EVERY run in the reference's order (ins_sim.py:387-392) unless there are
        more than ``max_summary_runs`` of them (default: 2048 entries, i.e. far beyond what the reference is ever run with);
        then the first ones by (algorithm, run number) are printed with a note -- ``sim.err_stats`` holds all of them."""
        if not self.sim_complete:
            print("Call Sim.run() to run the simulaltion first.")
            return None
        data_saved = []
        if data_dir is not None:
            data_dir = self._check_data_dir(data_dir)
            data_saved = self.dmgr.save_data(data_dir, max_runs=max_saved_runs)
        if gen_kml is True:
            self.dmgr.save_kml_files(data_dir)
        self._summary(data_dir, data_saved, err_stats_start, extra_opt, max_summary_runs)
        self.sim_results = True
        return self.dmgr.available

    def _summary(self, data_dir, data_saved, err_stats_start=0, extra_opt='', max_summary_runs=None):
        """Same text as Sim.__summary (ins_sim.py:339-413)."""
        d = self.dmgr
        line = '\n------------------------------------------------------------\n'
        s = line
        s += d.fs.description + ': [' + d.fs.name + '] = ' + str(d.fs.data) + ' ' + d.fs.units[0] + '\n'
        s += d.ref_frame.description + ': ' + str(d.ref_frame.data) + '\n'
        s += 'Simulation time duration: ' + str(len(d.time.data) / d.fs.data) + ' s' + '\n'
        s += 'Simulation runs: ' + str(self.sim_count) + '\n'
        if data_dir is not None:
            s += line + 'Simulation results are saved to ' + data_dir + '\n' + 'The following results are saved:\n'
            for i in data_saved:
                s += '\t' + i + ': ' + d.get_data_all(i).description + '\n'
        header = False
        self.err_stats = {}
        for data_name, kind in self.interested_error.items():
            if data_name not in d.available:
                continue
            st = d.get_error_stats(data_name, err_stats_start=err_stats_start, angle=(kind == 'angle'),
                                   use_output_units=True, extra_opt=extra_opt)
            if st is None:
                continue
            self.err_stats[data_name] = st
            if not header:
                header = True
                s += line + 'The following are error statistics.'
            s += '\n-----------statistics for ' + d.get_data_all(data_name).description + \
                 ' (in units of ' + st['units'] + ')\n'
            if hasattr(st['max'], 'keys'):
                limit = 2048 if max_summary_runs is None else int(max_summary_runs)
                total = len(st['max'])
                if total <= limit:
                    keys = sorted(st['max'].keys())      # the reference's (lexicographic) order
                elif hasattr(st['max'], 'first_keys'):   # truncating: first runs of every algorithm by run NUMBER,
                    keys = st['max'].first_keys(limit)   # made directly (10^5 keys are not built to print 2048 of them)
                else:
                    def run_order(k):
                        name, _, num = str(k).rpartition('_')
                        return (name, int(num)) if num.isdigit() else (str(k), -1)
                    keys = sorted(st['max'].keys(), key=run_order)[:limit]
                fast = sim_data.default_print_options()
                rows = []
                for k in keys:
                    rows.append('\tSimulation run ' + str(k) + ':\n'
                                '\t\t--Max error: ' + sim_data.vec_str(st['max'][k], fast) + '\n'
                                '\t\t--Avg error: ' + sim_data.vec_str(st['avg'][k], fast) + '\n'
                                '\t\t--Std of error: ' + sim_data.vec_str(st['std'][k], fast) + '\n')
                s += ''.join(rows)
                if total > limit:
                    s += '\t... %d more runs: sim.err_stats[%r]\n' % (total - limit, data_name)
            else:       # one vector per statistic (end-point mode): the same writer, a quarter of numpy's array printer
                fast = sim_data.default_print_options()
                s += '\t--Max error: ' + sim_data.vec_str(st['max'], fast) + '\n'
                s += '\t--Avg error: ' + sim_data.vec_str(st['avg'], fast) + '\n'
                s += '\t--Std of error: ' + sim_data.vec_str(st['std'], fast) + '\n'
        self.sum += s
        if self._dist()[0] == 0:
            print(self.sum)
            devs = getattr(self.mc, 'devices', None) if self.mc is not None else None
            if devs and len(devs) > 1:          # after the reference's text, never inside it (summary.txt stays the reference's)
                print('The runs were spread over %d devices of this process: %s' % (len(devs), ', '.join('cuda:%d' % k for k in devs)))
            if data_dir is not None:
                try:
                    with open(data_dir + '//summary.txt', 'w') as f:
                        f.write(self.sum + '\n')
                except Exception:
                    raise IOError('Unable to save summary to %s.' % data_dir)

    def plot(self, what_to_plot, sim_idx=None, opt=None, extra_opt=''):
        """Plotting is outside the accelerated path; the call is accepted (the reference's demo scripts end with it) and
        says where the data are instead of drawing them."""
        print('plot(%s): not drawn by this package -- the series are in sim.dmgr.<name>.data (per-run views) and in the '
              'CSV files of results(data_dir).' % (what_to_plot,))

    def get_names_of_available_data(self):
        return self.dmgr.available

    def get_data(self, data_names):
        return self.dmgr.get_data(data_names).copy()

    def get_data_properties(self, data_name):
        return self.dmgr.get_data_properties(data_name)

    # ------------------------------------------------------------------------------------ helpers
    def _parse_env(self, env):
        """Sim.__parse_env (ins_sim.py:642-701): one entry of the env dict -> the vib_def dict acc_gen / gyro_gen take.
        '[x y z]<g|d|>-random' (1 sigma) or '[x y z]<g|d|>-<f>Hz-sinusoidal' (peak); 'g' = 9.8 m/s^2, 'd' = deg/s;
        an (n,4) array [freq, x, y, z] is a single-sided PSD, cut at fs/2."""
        if env is None:
            return None
        if isinstance(env, np.ndarray):
            if env.ndim == 2 and env.shape[1] == 4:
                rows, half_fs = env.shape[0], 0.5 * self.fs[0]
                if env[-1, 0] > half_fs:
                    rows = np.where(env[:, 0] > half_fs)[0][0]
                return {'type': 'psd', 'freq': env[:rows, 0], 'x': env[:rows, 1], 'y': env[:rows, 2], 'z': env[:rows, 3]}
            raise TypeError('env should be of size (n,2)')
        if not isinstance(env, str):
            raise TypeError('env should be a string or a numpy array of size (n,2)')
        text, vib = env.lower(), {}
        if 'random' in text:
            vib['type'] = 'random'
            text = text.replace('-random', '')
        elif 'sinusoidal' in text:
            vib['type'] = 'sinusoidal'
            text = text.replace('-sinusoidal', '')
            if text[-2:] != 'hz':
                raise ValueError('env = \'%s\' is not valid (No vib freq).' % text)
            try:
                mark = text.find('-')
                vib['freq'] = math.fabs(float(text[mark + 1:-2]))
                text = text[:mark]
            except Exception:
                raise ValueError('env = \'%s\' is not valid (invalid vib freq).' % text)
        else:
            raise ValueError('env = \'%s\' is not valid.' % text)
        unit = 1.0                              # 1 sigma (random) or peak (sinusoidal)
        if text[-1] == 'g':                     # accelerations in g
            unit, text = 9.8, text[:-1]
        elif text[-1] == 'd':                   # angular rates in deg/s
            unit, text = attitude.D2R, text[:-1]
        try:
            amp = unit * np.array(text[1:-1].split(' '), dtype='float64')
            vib['x'], vib['y'], vib['z'] = amp[0], amp[1], amp[2]
        except Exception:
            raise ValueError('Cannot convert \'%s\' to float' % text[1:-1].split(' '))
        return vib

    @staticmethod
    def _parse_mode(mode):
        """Sim.__parse_mode (ins_sim.py:612-640)."""
        if mode is None or isinstance(mode, str):
            return high_mobility
        if isinstance(mode, np.ndarray):
            if mode.shape != (3,):
                raise TypeError('mode should be of size (3,)')
            m = mode.astype(np.float64)
            m[1] *= attitude.D2R
            m[2] *= attitude.D2R
            return m
        raise TypeError('mode should be a string or a numpy array of size (3,)')

    @staticmethod
    def _check_data_dir(data_dir):
        """Sim.__check_data_dir (ins_sim.py:703-727)."""
        if data_dir == '':
            data_dir = os.path.join(os.path.abspath('.//demo_saved_data//'),
                                    time.strftime('%Y-%m-%d-%H-%M-%S', time.localtime()))
        data_dir = os.path.abspath(data_dir)
        if not os.path.exists(data_dir):
            try:
                os.makedirs(data_dir)
            except Exception:
                raise IOError('Cannot create dir: %s.' % data_dir)
        return data_dir
