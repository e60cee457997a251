"""Named data series with units, interface-compatible with the reference's ``Sim_data``
(gnss_ins_sim/sim/sim_data.py:13-260): ``name/description/units/output_units/legend/data``,
``add_data``, ``save_to_file`` (same CSV header + np.savetxt format) and ``convert_unit``.

``data`` may be a scalar, an ndarray, a dict of arrays, or -- for Monte-Carlo series that live on the
GPU -- a ``McSeries``: a read-only mapping run-key -> (n,k) array that pulls a run out of HBM on demand.
"""
from collections import OrderedDict
from collections.abc import Mapping

import numpy as np

from ..attitude import attitude

_SCALES = {('deg', 'rad'): attitude.D2R, ('deg/s', 'rad/s'): attitude.D2R, ('deg/hr', 'rad/s'): attitude.D2R / 3600.0,
           ('rad', 'deg'): 1.0 / attitude.D2R, ('rad/s', 'deg/s'): 1.0 / attitude.D2R,
           ('rad/s', 'deg/hr'): 3600.0 / attitude.D2R}


def unit_conversion_scale(src_unit, dst_unit):
    """sim_data.unit_conversion_scale (sim_data.py:208-233)."""
    scale = np.ones(len(dst_unit))
    for i, (s, d) in enumerate(zip(src_unit, dst_unit)):
        if (s, d) in _SCALES:
            scale[i] = _SCALES[(s, d)]
        elif s != d:
            print('Cannot convert unit from %s in %s to %s.' % (s, src_unit, d))
    return scale


def convert_unit_ndarray_scalar(x, scale):
    """sim_data.convert_unit_ndarray_scalar (sim_data.py:235-260); never modifies its input."""
    m = scale.shape[0]
    if isinstance(x, np.ndarray):
        if x.ndim == 2:
            k = min(m, x.shape[1])
            x = x.copy()
            x[:, :k] = x[:, :k] * scale[:k]
            return x
        if x.ndim == 1:
            return x * scale if len(x) == m else x * scale[0]
        raise ValueError('Input x should be a scalar, 1D or 2D array, ndim = %s' % x.ndim)
    if isinstance(x, (int, float, np.floating, np.integer)):
        return x * scale[0]
    raise ValueError('Input x should be a scalar, 1D or 2D array')


def convert_unit(data, src_unit, dst_unit):
    """sim_data.convert_unit (sim_data.py:187-206)."""
    scale = unit_conversion_scale(src_unit, dst_unit)
    if isinstance(data, Mapping):
        return {k: convert_unit_ndarray_scalar(data[k], scale) for k in data}
    return convert_unit_ndarray_scalar(data, scale)


class McSeries(Mapping):
    """Read-only mapping  key -> ndarray  over a Monte-Carlo series resident in GPU memory.

    ``fetch(list_of_positions) -> (k, n, ncomp) ndarray`` pulls runs out of the device buffer
    (ginsim_gather_runs).  Keys are produced by ``key_of(position)``: the integer run id for sensor
    series (dmgr.accel.data[i], ins_sim.py:493) or '<algo>_<run>' for algorithm outputs
    (ins_algo_manager.py:95).  A small LRU keeps recently used runs on the host.
    """

    def __init__(self, count, fetch, key_of=None, pos_of=None, squeeze=False, cache=64):
        self._count, self._fetch, self._squeeze = int(count), fetch, squeeze
        self._key_of = key_of or (lambda i: i)
        self._pos_of = pos_of or (lambda k: k if isinstance(k, (int, np.integer)) and 0 <= k < self._count else None)
        self._cache, self._cap = OrderedDict(), cache

    def __len__(self):
        return self._count

    def __iter__(self):
        return (self._key_of(i) for i in range(self._count))

    def __contains__(self, key):
        try:
            return self._pos_of(key) is not None
        except Exception:
            return False

    def __getitem__(self, key):
        if key in self._cache:
            self._cache.move_to_end(key)
            return self._cache[key]
        pos = self._pos_of(key) if self.__contains__(key) else None
        if pos is None:
            raise KeyError(key)
        a = self._fetch([pos])[0]
        if self._squeeze:
            a = a[:, 0]
        self._cache[key] = a
        if len(self._cache) > self._cap:
            self._cache.popitem(last=False)
        return a

    def copy(self):
        return self

    def take(self, positions):
        """Several runs at once: (k, n, ncomp)."""
        a = self._fetch(list(positions))
        return a[:, :, 0] if self._squeeze else a


_DEFAULT_PRINT = {'precision': 8, 'floatmode': 'maxprec', 'suppress': False, 'sign': '-', 'legacy': False}


def default_print_options():
    """True when numpy's print options that decide how a float vector is written are the defaults."""
    o = np.get_printoptions()
    return all(o.get(k) == v for k, v in _DEFAULT_PRINT.items()) and o.get('formatter') is None


def vec_str(a, fast=True):
    """str(a) for a short 1-D float64 vector -- the text the reference's summary prints for every statistic -- four times
    faster than numpy's array printer: the same two passes over the same dragon4 calls (numpy/_core/arrayprint.py,
    FloatingFormat: exponent form when max >= 1e8, min < 1e-4 or max / min > 1000; common integer and fraction widths),
    without building a formatter object per vector.  Only vectors of up to THREE elements (every statistic of the summary): a
    longer one can exceed numpy's line width of 75 and is then wrapped, which this writer does not do.  Anything else
    (non-finite values, other shapes or types, changed print options: pass fast=default_print_options()) goes to str().
    tests/test_host_cpu.py compares the two on 40 000 vectors of sizes 1-3 and checks that sizes 4-8 take the str() path."""
    if not (fast and isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 1 and 0 < a.size <= 3):
        return str(a)
    vals = a.tolist()
    nz = [abs(v) for v in vals if v != 0.0]
    for v in vals:
        if v != v or v in (float('inf'), float('-inf')):
            return str(a)
    exp = False
    if nz:
        mx, mn = max(nz), min(nz)
        exp = mx >= 1.e8 or mn < 0.0001 or mx / mn > 1000.
    # one dragon4 call per value: the second pass of numpy's printer writes the same digits again and only pads them
    if exp:
        # ... the integer part to the common width on the left, the fraction with zeros and the exponent's digits with leading zeros
        fs = np.format_float_scientific
        parts = []
        for v in vals:
            fr, _, ex = fs(v, precision=8, unique=True, trim='.', sign=False).partition('e')
            ip, _, fpart = fr.partition('.')
            parts.append((ip, fpart, ex))
        pl = max(len(q[0]) for q in parts)
        pr = max(len(q[1]) for q in parts)
        es = max(len(q[2]) - 1 for q in parts)
        out = [q[0].rjust(pl) + '.' + q[1].ljust(pr, '0') + 'e' + q[2][0] + q[2][1:].rjust(es, '0') for q in parts]
    else:
        # ... the integer part to the common width on the left, the fraction with blanks on the right (the point stays: trim='.')
        fp = np.format_float_positional
        parts = [fp(v, precision=8, fractional=True, unique=True, trim='.', sign=False).partition('.') for v in vals]
        pl = max(len(q[0]) for q in parts)
        pr = max(len(q[2]) for q in parts)
        out = [q[0].rjust(pl) + '.' + q[2].ljust(pr) for q in parts]
    return '[' + ' '.join(out) + ']'


class RunStats(Mapping):
    """{'<algo>_<run>': (k,) statistics} over per-algorithm (runs, k) arrays: the per-run dicts of
    InsDataMgr.__process_error_stats (ins_data_manager.py:761-795) without building 10^5 dict entries up front."""

    def __init__(self, parts):
        self.parts = [(name, int(first), np.asarray(a)) for name, first, a in parts]      # (algo name, first run id, (runs, k))

    def __len__(self):
        return sum(a.shape[0] for _, _, a in self.parts)

    def __iter__(self):
        for name, first, a in self.parts:
            for i in range(a.shape[0]):
                yield name + '_' + str(first + i)

    def __getitem__(self, key):
        nm, _, r = str(key).rpartition('_')
        if r.isdigit():
            for name, first, a in self.parts:
                if name == nm and first <= int(r) < first + a.shape[0]:
                    return a[int(r) - first]
        raise KeyError(key)

    def first_keys(self, limit):
        """The first `limit` keys in (algorithm name, run number) order, without making the others."""
        out = []
        for name, first, a in sorted(self.parts, key=lambda p: (p[0], p[1])):
            take = min(a.shape[0], limit - len(out))
            out.extend(name + '_' + str(first + i) for i in range(take))
            if len(out) >= limit:
                break
        return out

    def scaled(self, scale):
        return RunStats([(name, first, a * scale[:a.shape[1]]) for name, first, a in self.parts])

    def update(self, other):
        """append host-computed entries ({key: array}) as single-run parts"""
        for k, v in other.items():
            nm, _, r = str(k).rpartition('_')
            self.parts.append((nm, int(r) if r.isdigit() else 0, np.asarray(v)[None]))


class ChainSeries(Mapping):
    """A device view (McSeries) followed by host arrays under further keys: the outputs of fused and hosted plugins
    that share an output name (e.g. 'pos' from FreeIntegration in the kernel and from a user's EKF on the host)."""

    def __init__(self, first, extra):
        self.first, self.extra = first, dict(extra)

    def __len__(self):
        return len(self.first) + len(self.extra)

    def __iter__(self):
        for k in self.first:
            yield k
        for k in self.extra:
            yield k

    def __contains__(self, key):
        return key in self.extra or key in self.first

    def __getitem__(self, key):
        return self.extra[key] if key in self.extra else self.first[key]

    def copy(self):
        return self


class DerivedSeries(Mapping):
    """key -> fn(key, source[key]), formed when the key is read (e.g. the error series of a run that lives on the GPU)."""

    def __init__(self, source, fn):
        self.source, self.fn = source, fn

    def __len__(self):
        return len(self.source)

    def __iter__(self):
        return iter(self.source)

    def __contains__(self, key):
        return key in self.source

    def __getitem__(self, key):
        return self.fn(key, self.source[key])

    def copy(self):
        return self


class Lazy(object):
    """A derived series that is computed when it is first read (``Sim_data.data``): e.g. the quaternion form of the true
    attitude (ins_sim.py:729-748), which the reference computes eagerly in every run() and no statistic uses."""

    def __init__(self, make):
        self.make = make


class Sim_data(object):
    @property
    def data(self):
        if isinstance(self._data, Lazy):
            self._data = self._data.make()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def __init__(self, name, description, units=None, output_units=None, plottable=True, logx=False, logy=False,
                 grid='on', legend=None):
        self.name, self.description = name, description
        self.units = [''] if units is None else list(units)
        if output_units is None:
            self.output_units = self.units
        else:
            self.output_units = list(output_units)
            while len(self.output_units) < len(self.units):
                self.output_units.append(self.units[len(self.output_units)])
            while len(self.units) < len(self.output_units):
                self.units.append(self.output_units[len(self.units)])
        self.plottable, self.logx, self.logy = plottable, logx, logy
        self.grid = grid if grid.lower() == 'off' else 'on'
        self.legend = legend
        self.data = {}

    def add_data(self, data, key=None, units=None):
        """Sim_data.add_data (sim_data.py:77-115)."""
        if units is not None:
            units = list(units)
            if len(units) != len(self.units):
                print(units)
                print(self.units)
                raise ValueError('Units are of different lengths.')
            if units != self.units:
                data = convert_unit(data, units, self.units)
        if key is None:
            self.data = data
        else:
            if not isinstance(self.data, dict):
                self.data = {}
            self.data[key] = data

    def _header(self, cols):
        if cols > 0:
            parts = []
            for i in range(cols):
                unit = ' (' + self.output_units[i] + ')' if i < len(self.output_units) else ''
                label = self.legend[i] if (self.legend is not None and cols == len(self.legend)) else self.name + '_' + str(i)
                parts.append(label + unit)
            return ','.join(parts)
        return self.name + (' (' + self.output_units[0] + ')' if len(self.output_units) > 0 else '')

    def save_to_file(self, data_dir, keys=None):
        """Sim_data.save_to_file (sim_data.py:117-165): '<name>[-<key>].csv', header 'legend (unit)',
        np.savetxt default %.18e.  ``keys`` limits which runs of a keyed series are written."""
        if isinstance(self.data, Mapping):
            for k in (self.data if keys is None else [k for k in keys if k in self.data]):
                a = np.asarray(self.data[k])
                cols = a.shape[1] if a.ndim > 1 else 0
                np.savetxt(data_dir + '//' + self.name + '-' + str(k) + '.csv',
                           convert_unit(a, self.units, self.output_units), header=self._header(cols),
                           delimiter=',', comments='')
        else:
            a = np.asarray(self.data)
            cols = a.shape[1] if a.ndim > 1 else 0
            np.savetxt(data_dir + '//' + self.name + '.csv', convert_unit(a, self.units, self.output_units)
                       if a.ndim > 0 else np.atleast_1d(a), header=self._header(cols), delimiter=',', comments='')

    def plot(self, *args, **kwargs):
        raise NotImplementedError('plotting is outside the accelerated hot path (SURVEY.md section 2, #17)')
