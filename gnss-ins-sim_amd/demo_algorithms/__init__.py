"""The plugins of the accelerated path (free_integration, free_integration_odo, allan_analysis) under the reference's package
name.  Any other ``demo_algorithms`` module (inclinometer_mahony, mag_calibrate, aceinna_ins ...) is outside the path: it is
imported from the reference checkout named by $GNSS_INS_SIM_REFERENCE when there is one (gnss_ins_sim/_reference.py)."""
from gnss_ins_sim import _reference as _reference

_reference.install()
