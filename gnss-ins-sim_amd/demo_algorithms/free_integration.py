"""IMU free integration (strapdown mechanisation) plugin -- same surface as the reference's
demo_algorithms/free_integration.py (FreeIntegration(ini_pos_vel_att, earth_rot=True), .input, .output,
.batch, .results, .run(set_of_input), .get_results(), .reset()), executed by the HIP kernel.

* ``run([ref_frame, fs, gyro(n,3), accel(n,3)])`` integrates ONE given sensor record on the GPU
  (ginsim_free_integration, the given-data entry point) and returns att/pos/vel exactly like the reference.
* inside ``Sim.run(N)`` the plugin is not called per run at all: ``mc_algo = 'free'`` tells Sim to integrate all
  N runs in the fused Monte-Carlo kernel.
There is no NumPy implementation behind this class; without a GPU ``run`` raises.
"""
import numpy as np


class FreeIntegration(object):
    mc_algo = 'free'

    def __init__(self, ini_pos_vel_att, earth_rot=True):
        self.input = ['ref_frame', 'fs', 'gyro', 'accel']
        self.output = ['att_euler', 'pos', 'vel']
        self.earth_rot = earth_rot
        self.batch = True
        self.results = None
        self.ref_frame = 1
        self.dt = 1.0
        self.att = self.pos = self.vel = self.vel_b = None
        ini = np.asarray(ini_pos_vel_att, dtype=np.float64)
        if ini.ndim == 1:                                   # free_integration.py:47-49
            self.set_of_inis = 1
            ini = ini.reshape((ini.shape[0], 1))
        elif ini.ndim == 2:                                 # :51-52
            self.set_of_inis = ini.shape[1]
        else:
            raise ValueError('Initial states should be a 1D or 2D numpy array, \
                              but the dimension is %s.' % ini.ndim)
        self.run_times = int(0)
        self.ini = ini
        self.r0, self.v0, self.att0 = ini[0:3], ini[3:6], ini[6:9]
        self.gravity = ini[9] if len(ini) > 9 else None

    def _inputs(self, set_of_input):
        return set_of_input[2], set_of_input[3], None

    def run(self, set_of_input):
        import ginsim
        first = self.run_times
        self.run_times += 1
        if set_of_input[0] == 0:                            # sticky, like free_integration.py:71-72
            self.ref_frame = 0
        self.dt = 1.0 / set_of_input[1]
        gyro, accel, odo = self._inputs(set_of_input)
        self.att, self.pos, self.vel = ginsim.free_integration_host(
            ginsim.default_context(), self.mc_algo, self.ref_frame, set_of_input[1], gyro, accel=accel, odo=odo,
            ini=self.ini, earth_rot=self.earth_rot, ini_first=first)
        self.results = [self.att, self.pos, self.vel]

    def get_results(self):
        return self.results

    def reset(self):
        pass
