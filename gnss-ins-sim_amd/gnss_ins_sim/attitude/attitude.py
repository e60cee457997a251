"""ZYX subset of the reference's attitude module (gnss_ins_sim/attitude/attitude.py), vectorised NumPy.

Host-side helpers for DERIVED data only (lazy att_quat, unit handling).  The per-timestep attitude
propagation of the hot path lives in csrc/ins_math.hpp and runs on the GPU; its one-step form is offered here under its
reference name (``euler_update_zyx``, native host code of libginsim) for hosted plugins that call it step by step.
Every other name of the reference's module (quat_update, dcm2quat, rot_x ...) is looked up in the reference checkout named by
$GNSS_INS_SIM_REFERENCE (module ``__getattr__`` -> gnss_ins_sim/_reference.py).
"""
import math

import numpy as np

D2R = math.pi / 180.0
R2D = 180.0 / math.pi
TWO_PI = 2.0 * math.pi
HALF_PI = 0.5 * math.pi


def euler2quat(angles, rot_seq='zyx'):
    """attitude.euler2quat (attitude.py:188-205) for 'zyx'; angles (...,3) -> (...,4), scalar first."""
    if rot_seq.lower() != 'zyx':
        raise NotImplementedError('only the ZYX sequence is on the accelerated path')
    a = np.asarray(angles, dtype=np.float64)
    c, s = np.cos(0.5 * a), np.sin(0.5 * a)
    return np.stack([c[..., 0] * c[..., 1] * c[..., 2] + s[..., 0] * s[..., 1] * s[..., 2],
                     c[..., 0] * c[..., 1] * s[..., 2] - s[..., 0] * s[..., 1] * c[..., 2],
                     c[..., 0] * s[..., 1] * c[..., 2] + s[..., 0] * c[..., 1] * s[..., 2],
                     s[..., 0] * c[..., 1] * c[..., 2] - c[..., 0] * s[..., 1] * s[..., 2]], axis=-1)


def euler2dcm(angles, rot_seq='zyx'):
    """attitude.euler2dcm (attitude.py:344-371) for 'zyx': n -> b."""
    if rot_seq.lower() != 'zyx':
        raise NotImplementedError('only the ZYX sequence is on the accelerated path')
    a = np.asarray(angles, dtype=np.float64)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c[1] * c[0], c[1] * s[0], -s[1]],
                     [s[2] * s[1] * c[0] - c[2] * s[0], s[2] * s[1] * s[0] + c[2] * c[0], c[1] * s[2]],
                     [s[1] * c[2] * c[0] + s[0] * s[2], s[1] * c[2] * s[0] - c[0] * s[2], c[1] * c[2]]])


def cross3(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def angle_range_pi(x):
    """attitude.angle_range_pi (attitude.py:799-812), scalar or array."""
    x = np.mod(x, TWO_PI)
    return np.where(x > math.pi, x - TWO_PI, x) if isinstance(x, np.ndarray) else (x - TWO_PI if x > math.pi else x)


def euler_update_zyx(x, w, dt):
    """attitude.euler_update_zyx (attitude.py:679-721): propagate the ZYX Euler angles x = [yaw, pitch, roll] by the body rate w
    over dt -- Euler-rate integration, pitch fold at +-pi/2, ONE +-2 pi wrap of yaw and roll.  Through the C ABI
    (ginsim_euler_update_zyx: native host code of libginsim; the kernels run the same update with cached trigonometry)."""
    import ginsim
    from ginsim._lib import lib, check, dptr
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(3))
    w = np.ascontiguousarray(np.asarray(w, dtype=np.float64).reshape(3))
    y = np.empty(3)
    check(lib.ginsim_euler_update_zyx(dptr(x), dptr(w), float(dt), dptr(y)))
    return y


def __getattr__(name):
    from .. import _reference
    return _reference.delegate(__name__, name)
