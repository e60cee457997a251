import os
import sys

import numpy as np
import pytest

# tests may load single files of the read-only reference checkout (geomag.py): never drop bytecode next to them
sys.dont_write_bytecode = True
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'gnss-ins-sim_amd')
GOLDEN = os.path.join(REPO, 'tests', 'golden')
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_built():
    """The shared libraries are git-ignored build products: a fresh checkout builds them on first use
    (hipcc cross-compiles gfx950 without a GPU; ~1 min).  Existing up-to-date libraries are left alone."""
    lib = os.path.join(PKG, 'lib', 'libginsim.so')
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location('ginsim_build', os.path.join(PKG, 'build.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    _ensure_built()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def golden_vibration(g):
    """(vib_def of the accelerometer, of the gyroscope) of a T3 golden made with Sim(env=...): the dicts the reference's own
    Sim.__parse_env produced (ins_sim.py:642-701), or None."""
    out = []
    for sensor in ('acc', 'gyro'):
        if 'vib_%s_type' % sensor not in g:
            out.append(None)
            continue
        amp = g['vib_%s_amp' % sensor]
        if str(g['vib_%s_type' % sensor]) == 'psd':      # fresh arrays, as they were BEFORE the reference's run (it halves a PSD given
            out.append({'type': 'psd', 'freq': g['vib_%s_freq' % sensor].copy(), 'x': amp[0].copy(), 'y': amp[1].copy(), 'z': amp[2].copy()})
            continue                                     # on the series' own grid in place: time_series_from_psd.py:44-49)
        v = {'type': str(g['vib_%s_type' % sensor]), 'x': float(amp[0]), 'y': float(amp[1]), 'z': float(amp[2])}
        if v['type'] == 'sinusoidal':
            v['freq'] = float(g['vib_%s_freq' % sensor])
        out.append(v)
    return tuple(out)


T3_VIB = ['t3_vib_random_rf1', 't3_vib_sin_rf0', 't3_vib_mixed_rf1']
# Sim(env=<(n, 4) PSD array>): interpolated + given on the grid (halved in place per run), an odd series length, a series tiled beyond 16384
T3_PSD = ['t3_vib_psd_rf1', 't3_vib_psd_odd_rf0', 't3_vib_psd_tiled_rf0']


@pytest.fixture(scope='session')
def golden():
    return load_golden


def ang_close(a, b, tol):
    """Angles compared modulo 2*pi (a 1-ulp difference at the +-pi wrap must not count)."""
    d = np.mod(np.asarray(a) - np.asarray(b) + np.pi, 2 * np.pi) - np.pi
    return np.max(np.abs(d)) <= tol if d.size else True


def assert_traj_close(att, pos, vel, g_att, g_pos, g_vel, rtol=1e-9, what=''):
    """fp64 trajectory tolerance of SURVEY section 8(c): |d| <= rtol * max(1, |x|), angles mod 2*pi.  ECEF-sized
    coordinates (ref_frame 1 positions, ~5e6 m) are held to an ABSOLUTE 2e-8 m (a few ulp) instead of the 5 mm the
    relative rule would allow."""
    assert ang_close(att, g_att, rtol * 4), what + ' att'
    for name, x, g in (('pos', pos, g_pos), ('vel', vel, g_vel)):
        tol = rtol * np.maximum(1.0, np.abs(g))
        tol = np.where(np.abs(g) > 1e5, np.minimum(tol, 2e-8), tol)
        bad = np.abs(x - g) - tol
        assert np.max(bad) <= 0.0, '%s %s: |d| exceeds the tolerance by %.3e' % (what, name, np.max(bad))
