"""Allan variance with the reference's function name and return convention
(gnss_ins_sim/allan/allan.py:18-59: ``allan_var(x, fs) -> (avar, tau)``), computed by the HIP kernels of
csrc/allan.hip through ginsim_allan.  No NumPy implementation behind it."""
import numpy as np


def allan_var(x, fs):
    import ginsim
    x = np.asarray(x, dtype=np.float64)
    avar, tau = ginsim.allan_var_host(ginsim.default_context(), x, fs)
    if tau.size == 0:
        return [], []            # allan.py:30-31
    return avar, tau
