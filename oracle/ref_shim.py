"""``np.random.randn`` shim that feeds the UNMODIFIED reference the engine's counter-based normals
(TEST ORACLE -- see oracle/__init__.py).

The reference resolves ``np.random.randn`` at call time, so replacing the attribute for the
duration of ``Sim.run`` injects our normals in the reference's own call order.  Call order per
Monte-Carlo run (Sim.__gen_data_from_pathgen, /root/reference/gnss_ins_sim/sim/ins_sim.py:490-506):

    acc_gen : bias_drift axis 0,1,2 -> randn(n,3) each, column i used (pathgen.py:588-590)
                                       or randn(n) when b_corr[i] is inf (pathgen.py:593)
              white noise            -> randn(n,3)            (pathgen.py:495)
    gyro_gen: same                                            (pathgen.py:537, 557)
    gps_gen : randn(m,3) position, randn(m,3) velocity        (pathgen.py:621-622)
    mag_gen : randn(n,3)                                      (pathgen.py:660)
    odo_gen : randn(n)                                        (pathgen.py:639)

Only available where /root/reference exists (the build container); never on the GPU box.
"""
import contextlib
import numpy as np

from . import philox


class RandnShim:
    def __init__(self, seed, n, acc_corr, gyro_corr, gps_m=0, mag=False, odo=False, first_run=0, vib_acc=None, vib_gyro=None):
        """vib_acc / vib_gyro: None | 'random' | 'sinusoidal' | ('psd', L) -- the vibration type Sim(env=...) gives each sensor: a
        'random' vibration draws randn(n) three times between the drift and the white-noise draws (pathgen.py:486-488, 548-550); a
        'psd' vibration randn(L) three times at the same place (:479-484, :541-546 -> time_series_from_psd.py:52; the status
        False path draws nothing: pass None); a 'sinusoidal' gyro vibration draws np.random.rand(1) three times (:553-555),
        served by ``rand``."""
        self.seed, self.n = seed, n
        self.vib = (vib_acc, vib_gyro)
        self._phase_k = 0
        self.acc_inf = np.isinf(np.asarray(acc_corr, dtype=np.float64))
        self.gyro_inf = np.isinf(np.asarray(gyro_corr, dtype=np.float64))
        self.gps_m, self.mag, self.odo = gps_m, mag, odo
        self.run = first_run
        self.queue = []

    def _plan(self):
        z = philox.imu_normals(self.seed, self.run, self.n)
        q = []
        for d, w, inf, vib, sensor in ((z['acc_d'], z['acc_w'], self.acc_inf, self.vib[0], 'acc'),
                                       (z['gyr_d'], z['gyr_w'], self.gyro_inf, self.vib[1], 'gyr')):
            for i in range(3):
                q.append(d[:, i].copy() if inf[i] else d.copy())
            if vib == 'random':
                v = philox.vib_normals(self.seed, self.run, self.n, sensor)
                q += [v[:, 0].copy(), v[:, 1].copy(), v[:, 2].copy()]
            elif isinstance(vib, tuple) and vib[0] == 'psd':        # randn(L) for x, y, z (time_series_from_psd.py:52)
                v = philox.vib_normals(self.seed, self.run, vib[1], sensor)
                q += [v[:, 0].copy(), v[:, 1].copy(), v[:, 2].copy()]
            q.append(w.copy())
        if self.gps_m:
            p, v = philox.gps_normals(self.seed, self.run, self.gps_m)
            q += [p, v]
        if self.mag:
            q.append(philox.mag_normals(self.seed, self.run, self.n))
        if self.odo:
            q.append(philox.odo_normals(self.seed, self.run, self.n))
        self.queue = q
        self.run += 1

    def __call__(self, *shape):
        if not self.queue:
            self._plan()
        out = self.queue.pop(0)
        if tuple(shape) != out.shape:
            raise AssertionError('reference asked randn%s, shim planned %s' % (shape, out.shape))
        return out


    def rand(self, *shape):
        """np.random.rand(1) of gyro_gen's sinusoidal vibration: the phase uniforms of the CURRENT run, x, y, z in turn."""
        if tuple(shape) != (1,) or self.vib[1] != 'sinusoidal' or self.run == 0:
            raise AssertionError('reference asked rand%s outside a sinusoidal gyro vibration' % (shape,))
        u = philox.vib_phase_uniforms(self.seed, self.run - 1, 'gyr')[self._phase_k]
        self._phase_k = (self._phase_k + 1) % 3
        return np.array([u])


@contextlib.contextmanager
def injected(shim):
    saved, saved_rand = np.random.randn, np.random.rand
    np.random.randn = shim
    np.random.rand = shim.rand
    try:
        yield shim
    finally:
        np.random.randn, np.random.rand = saved, saved_rand
