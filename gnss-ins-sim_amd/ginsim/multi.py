"""Several GPUs from ONE process: a ``DeviceSet`` holds one ``Context`` per entry of a device list and a ``JobSet`` is
one Monte-Carlo batch spread over them.

The reference's Monte-Carlo loop (gnss_ins_sim/sim/ins_sim.py:490-506: ``for i in range(self.sim_count)``) is what is being
sharded: device d of D takes the contiguous global runs ``shard(runs, D, d)`` -- the same split the one-process-per-GPU form
uses (ginsim.distributed) -- and because the Philox counter carries the GLOBAL run id, every run is bit-identical to the
same run of a single launch, whatever D is.  The per-device end-point records are folded with the library's Chan merge
(``ginsim_stats_merge``) in device order.  No torch, no launcher, no collective: the records are 28 doubles per device and
come back through each context's own pinned slot.

Threads: the C ABI is re-entrant per context handle (every entry point selects its device, the error string is
thread-local) and ctypes releases the GIL for the duration of a call, so each context is driven by its own Python thread
(``DeviceSet.each``): uploads, launches, synchronisation and read-back of D devices overlap.  The same device may appear
more than once in the list (``devices=[0, 0]``): two contexts, two streams, one GPU -- which is how the one-GPU test box
executes this path.

A ``JobSet`` answers the calls ``Sim`` and the plugins make on a ``MonteCarloJob`` (statistics, per-run series, Allan
analysis), routing a run id to the device that holds it.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .engine import Context, MonteCarloJob, AuxSensorJob, StatsResult, device_count
from .distributed import shard


def parse_devices(devices):
    """'all' -> every visible device; an int -> that device; a sequence of ids (repeats allowed) -> as given."""
    n = device_count()
    if isinstance(devices, str):
        if devices != 'all':
            raise ValueError("devices: 'all', a device index or a list of device indices")
        if n < 1:
            raise ValueError('devices=%r: no HIP device visible (the engine has no CPU fallback)' % (devices,))
        return list(range(n))
    if isinstance(devices, (int, np.integer)):
        devices = [int(devices)]
    ids = [int(d) for d in devices]
    if not ids:
        raise ValueError('devices: empty list')
    for d in ids:
        if not 0 <= d < max(n, 1):
            raise ValueError('devices: device %d out of range [0, %d)' % (d, n))
    return ids


class DeviceSet(object):
    """One Context per entry of `devices`, each driven by its own thread."""

    def __init__(self, devices='all'):
        self.devices = parse_devices(devices)
        self.contexts = [Context(d) for d in self.devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.contexts), thread_name_prefix='ginsim-dev')

    def __len__(self):
        return len(self.contexts)

    def each(self, fn, items=None):
        """fn(k, context_k[, items[k]]) on every context at the same time, one thread per context; results in context order.
        The first exception (in context order) is re-raised after every thread has finished."""
        ks = range(len(self.contexts))
        if items is None:
            futs = [self._pool.submit(fn, k, self.contexts[k]) for k in ks]
        else:
            futs = [self._pool.submit(fn, k, self.contexts[k], items[k]) for k in ks]
        out, err = [], None
        for f in futs:
            try:
                out.append(f.result())
            except BaseException as e:      # noqa: BLE001 -- collected, re-raised below
                out.append(None)
                err = err or e
        if err is not None:
            raise err
        return out

    def sync(self):
        self.each(lambda k, c: c.sync())

    def close(self):
        for c in self.contexts:
            c.close()
        self._pool.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Parts(object):
    """Common routing of a set of per-device jobs over contiguous run ranges."""

    def _init_parts(self, devset, runs):
        self.devset, self.runs = devset, int(runs)
        D = len(devset)
        self.ranges = [shard(self.runs, D, d) for d in range(D)]                  # (first local run, count) per device
        self._bounds = np.array([f for f, _ in self.ranges] + [self.runs], dtype=np.int64)

    def _each_part(self, fn):
        """fn(part) on every device that holds runs, one thread per context; results in device order (None where empty)."""
        return self.devset.each(lambda k, c: None if self.parts[k] is None else fn(self.parts[k]))

    def _route(self, run_ids, fetch):
        """fetch(part, local_ids) -> (k, ...) arrays, reassembled in the order of run_ids."""
        ids = np.asarray(run_ids, dtype=np.int64).reshape(-1)
        if ids.size and (ids.min() < 0 or ids.max() >= self.runs):
            raise ValueError('run id out of range [0, %d)' % self.runs)
        if ids.size == 0:                       # nothing asked for: the (empty) answer of any device that holds runs
            return fetch(next(j for j in self.parts if j is not None), ids)
        owner = np.searchsorted(self._bounds, ids, side='right') - 1
        chunks = {}
        for d in np.unique(owner):
            sel = np.where(owner == d)[0]
            chunks[int(d)] = (sel, ids[sel] - self.ranges[int(d)][0])
        got = self.devset.each(lambda k, c: fetch(self.parts[k], chunks[k][1]) if k in chunks else None)
        first = next(g for g in got if g is not None)
        multi = isinstance(first, tuple)
        shapes = first if multi else (first,)
        outs = [np.empty((ids.size,) + s.shape[1:], dtype=s.dtype) for s in shapes]
        for d, (sel, _) in chunks.items():
            res = got[d] if multi else (got[d],)
            for o, a in zip(outs, res):
                o[sel] = a
        return tuple(outs) if multi else outs[0]


class JobSet(_Parts):
    """ginsim.MonteCarloJob over a DeviceSet: the same constructor arguments; `runs`, `run_offset` and `ini_first` describe the
    whole batch and every device gets its contiguous share.  Jobs that read device-resident input series (``given=``) belong
    to one context and are not spread."""

    def __init__(self, devset, fs, ref_frame, truth, accel_err, gyro_err, ini, runs, run_offset=0, ini_first=0, **kw):
        if kw.get('given') is not None:
            raise ValueError('given sensors live on one device: use a MonteCarloJob on that context')
        self._init_parts(devset, runs)

        def make(k, ctx):
            first, count = self.ranges[k]
            if count == 0:
                return None
            return MonteCarloJob(ctx, fs, ref_frame, truth, accel_err, gyro_err, ini, runs=count,
                                 run_offset=int(run_offset) + first, ini_first=int(ini_first) + first, **kw)
        self.parts = devset.each(make)
        p = next(j for j in self.parts if j is not None)
        self.algos, self.n, self.precision = p.algos, p.n, p.precision
        self.keep_sensors, self.keep_traj, self.want_odo = p.keep_sensors, p.keep_traj, p.want_odo
        self.proc_first, self.proc_ned, self.end_ned = p.proc_first, p.proc_ned, p.end_ned
        self.sensor_layout = p.sensor_layout
        self.params = p.params                  # of the first device's share (fs, ref_frame, ... are common)

    # ---- execution
    def launch(self):
        self._each_part(lambda j: j.launch())

    def run(self):
        self._each_part(lambda j: j.run())
        return self

    def kernel_name(self):
        return next(j for j in self.parts if j is not None).kernel_name()

    def bytes_written(self):
        return sum(j.bytes_written() for j in self.parts if j is not None)

    def release(self):
        self._each_part(lambda j: j.release())

    def placement(self):
        """MonteCarloJob.placement of every part, device order."""
        return self._each_part(lambda j: j.placement())

    # ---- statistics: per-device records folded with the library's Chan merge, device order
    def _merged(self, fn):
        return StatsResult.merge([s.pack() for s in self._each_part(fn) if s is not None])

    def stats(self, algo, ned=False):
        return self._merged(lambda j: j.stats(algo, ned=ned))

    def stats_from_traj(self, algo, pos_ned=False):
        return self._merged(lambda j: j.stats_from_traj(algo, pos_ned=pos_ned))

    def part_stats(self, algo, ned=False):
        """The unmerged per-device records (device order; None where a device holds no runs)."""
        return self._each_part(lambda j: j.stats(algo, ned=ned))

    def _concat(self, fn):
        return np.concatenate([a for a in self._each_part(fn) if a is not None], axis=0)

    def end_errors(self, algo, ned=False):
        return self._concat(lambda j: j.end_errors(algo, ned=ned))

    def process_stats_online(self, algo):
        return self._concat(lambda j: j.process_stats_online(algo))

    def process_stats(self, algo, first_sample=0, pos_ned=False):
        return self._concat(lambda j: j.process_stats(algo, first_sample, pos_ned=pos_ned))

    # ---- per-run series
    def sensors(self, name, run_ids):
        return self._route(run_ids, lambda j, ids: j.sensors(name, ids))

    def trajectories(self, algo, run_ids, displacement=False):
        return self._route(run_ids, lambda j, ids: j.trajectories(algo, ids, displacement=displacement))

    def allan(self, fs=None, names=('accel', 'gyro')):
        """MonteCarloJob.allan on every device at the same time: (tau, {name: (runs, ntau, 3)})."""
        got = [g for g in self._each_part(lambda j: j.allan(fs, names)) if g is not None]
        return got[0][0], {nm: np.concatenate([g[1][nm] for g in got], axis=0) for nm in names}

    def buffer(self, name):
        raise ValueError('a JobSet spans several devices: take the buffer from one of its parts')


class AuxJobSet(_Parts):
    """ginsim.AuxSensorJob (GPS / magnetometer series) over a DeviceSet."""

    def __init__(self, devset, runs, seed=0, run_offset=0, **kw):
        self._init_parts(devset, runs)

        def make(k, ctx):
            first, count = self.ranges[k]
            return None if count == 0 else AuxSensorJob(ctx, count, seed=seed, run_offset=int(run_offset) + first, **kw)
        self.parts = devset.each(make)
        p = next(j for j in self.parts if j is not None)
        self.m, self.n = p.m, p.n

    def run(self):
        self._each_part(lambda j: j.run())
        return self

    def series(self, name, run_ids):
        return self._route(run_ids, lambda j, ids: j.series(name, ids))

    def release(self):
        self._each_part(lambda j: j.release())
