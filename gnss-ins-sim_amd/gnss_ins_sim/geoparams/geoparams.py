"""WGS-84 helpers with the reference's names (gnss_ins_sim/geoparams/geoparams.py), host side.
The per-step Earth model of the hot path is csrc/ins_math.hpp::geo_param (device).  Names of the reference's module that are
not here (earth_radius, ecef2lla ...) fall through to a reference checkout named by $GNSS_INS_SIM_REFERENCE."""
import math

import numpy as np

GM = 3.986004418e14
Re = 6378137
FLATTENING = 1 / 298.257223563
ECCENTRICITY = 0.0818191908426215
E_SQR = ECCENTRICITY ** 2
W_IE = 7292115e-11


def geo_param(pos):
    """geoparams.geo_param (geoparams.py:25-53): (rm, rn, g, sl, cl, w_ie) at [lat, lon, alt]."""
    sl, cl = math.sin(pos[0]), math.cos(pos[0])
    s2, h = sl * sl, pos[2]
    w = math.sqrt(1.0 - E_SQR * s2)
    rm = (Re * (1 - E_SQR)) / (w * (1.0 - E_SQR * s2))
    rn = Re / w
    g = 9.7803253359 * (1 + 0.00193185265241 * s2) / w
    g *= 1.0 - (2.0 / Re) * (1.0 + FLATTENING + 0.00344978650684 - 2.0 * FLATTENING * s2) * h + 3.0 * h * h / Re / Re
    return rm, rn, g, sl, cl, W_IE


def lla2ecef(lla):
    """geoparams.lla2ecef / lla2ecef_batch (geoparams.py:70-113): (3,) or (n,3)."""
    lla = np.asarray(lla, dtype=np.float64)
    sl, cl = np.sin(lla[..., 0]), np.cos(lla[..., 0])
    r = Re / np.sqrt(1.0 - E_SQR * sl * sl)
    rho = (r + lla[..., 2]) * cl
    return np.stack([rho * np.cos(lla[..., 1]), rho * np.sin(lla[..., 1]), (r * (1.0 - E_SQR) + lla[..., 2]) * sl], -1)


lla2ecef_batch = lla2ecef


def reference_geomag_n(lat, lon, alt, when=None, root=None):
    """The geomagnetic field [uT] in the N frame at (lat, lon [rad], alt [m]) exactly as pathgen.path_gen obtains it
    (pathgen.py:164-168): ``geomag.GeoMag("WMM.COF").GeoMag(lat_deg, lon_deg, alt)`` -> (bx, by, bz) / 1000.  The World Magnetic
    Model itself is outside the accelerated path (SURVEY #14: "reuse on host"), so this only LOCATES the reference's own
    ``gnss_ins_sim/geoparams/geomag.py`` (+ WMM.COF) and loads that one file under a private module name (the package name
    itself is taken by this drop-in), without writing bytecode next to it.  Where it looks: `root`, else $GNSS_INS_SIM_REFERENCE
    -- an explicit statement by the caller, nothing is picked up from sys.path or the current directory.  Returns None when
    there is no such checkout.
    `when`: a datetime.date; the default is today -- the reference's own import-time default (geomag.py:23), which makes 9-axis
    outputs differ from one day to the next; the date used is printed once so that a run can be repeated (geo_mag_date=...)."""
    import datetime
    import importlib.util
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = root or os.environ.get('GNSS_INS_SIM_REFERENCE')
    if not root:
        return None
    d = os.path.join(root, 'gnss_ins_sim', 'geoparams')
    f = os.path.join(d, 'geomag.py')
    if os.path.abspath(d) == here or not (os.path.isfile(f) and os.path.isfile(os.path.join(d, 'WMM.COF'))):
        return None
    keep = sys.dont_write_bytecode
    sys.dont_write_bytecode = True              # the checkout may be read-only and is not ours to write into
    try:
        spec = importlib.util.spec_from_file_location('_ginsim_reference_geomag', f)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = keep
    gm = mod.GeoMag('WMM.COF')
    d2r = math.pi / 180                          # pos_n[0] / D2R, pathgen.py:166
    if when is None:
        when = datetime.date.today()
        print('geomagnetic field evaluated by %s for %s (pass geo_mag_date to repeat this run another day)' % (f, when.isoformat()))
    r = gm.GeoMag(lat / d2r, lon / d2r, alt, when)
    return np.array([r.bx, r.by, r.bz]) / 1000.0


def __getattr__(name):
    from .. import _reference
    return _reference.delegate(__name__, name)
