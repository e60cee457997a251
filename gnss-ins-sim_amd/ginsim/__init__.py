"""ginsim -- MI355X-native Monte-Carlo strapdown-INS engine (host side of libginsim.so).

Importing this package loads the HIP library through ctypes; it raises if the library has not been
built.  There is no CPU implementation behind it.
"""
from ._lib import lib, LIB_PATH, EXPORTS, GinsimError, GinsimOutOfMemory, PlacedUnavailable, ALGO_FREE, ALGO_ODO          # noqa: F401
from .engine import (Context, DeviceBuffer, MonteCarloJob, AuxSensorJob, StatsResult, device_count, pathgen, pinned_empty, vibration, psd_amplitudes,  # noqa: F401
                     sensor_model, ini_table, free_integration_host, rng_normals, normal_transform, default_context, allan_var, allan_var_host)
