"""Odometer-aided free integration plugin -- same surface as the reference's
demo_algorithms/free_integration_odo.py (attitude from the gyro, body velocity = [odo, 0, 0]), executed by
the HIP kernel (``mc_algo = 'odo'``).  See free_integration.py in this directory."""
from .free_integration import FreeIntegration as _Base


class FreeIntegration(_Base):
    mc_algo = 'odo'

    def __init__(self, ini_pos_vel_att, earth_rot=True):
        _Base.__init__(self, ini_pos_vel_att, earth_rot)
        self.input = ['ref_frame', 'fs', 'gyro', 'odo']

    def _inputs(self, set_of_input):
        return set_of_input[2], None, set_of_input[3]
