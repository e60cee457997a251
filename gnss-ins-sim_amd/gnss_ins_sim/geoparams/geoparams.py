"""WGS-84 helpers with the reference's names (gnss_ins_sim/geoparams/geoparams.py), host side.
The per-step Earth model of the hot path is csrc/ins_math.hpp::geo_param (device)."""
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
