"""The leaves of the hot path under their reference names (golden ``leaves.npz``: the unmodified reference evaluated on seeded
inputs, tests/golden/make_golden.py::leaves_case) and the fall-through of everything that is OFF the path to a reference
checkout named by $GNSS_INS_SIM_REFERENCE (gnss_ins_sim/_reference.py).

    attitude.euler_update_zyx          attitude.py:679-721
    pathgen.calc_true_sensor_output    pathgen.py:331-411
    pathgen.parse_motion_def           pathgen.py:413-439
    pathgen.bias_drift                 pathgen.py:565-594     (GPU)
    InsDataMgr.array_error / calc_data_err   ins_data_manager.py:454-553
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, REPO, PKG

REF = '/root/reference'
need_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'gnss_ins_sim')),
                              reason='no reference checkout at %s (build container only)' % REF)


def test_euler_update_zyx_is_the_references():
    from gnss_ins_sim.attitude import attitude
    g = load_golden('leaves')
    y = np.array([attitude.euler_update_zyx(g['eu_x'][i], g['eu_w'][i], g['eu_dt'][i]) for i in range(g['eu_x'].shape[0])])
    # same libm, same order of operations: equal to the last bit, folds and wraps included
    np.testing.assert_array_equal(y, g['eu_y'])
    folded = np.abs(g['eu_x'][:, 1] + 0) > 1.55
    assert folded.sum() >= 50 and np.any(np.abs(y[:, 0] - g['eu_x'][:, 0]) > 3.0)       # the fold / wrap branches were taken
    x = g['eu_x'][0].copy()
    attitude.euler_update_zyx(x, g['eu_w'][0], 0.01)
    np.testing.assert_array_equal(x, g['eu_x'][0])                                        # the input is not modified


@pytest.mark.parametrize('rf', [0, 1])
def test_calc_true_sensor_output_is_the_references(rf):
    from gnss_ins_sim.pathgen import pathgen
    g = load_golden('leaves')
    t = lambda k: g['ts%d_%s' % (rf, k)]
    for i in range(t('pos').shape[0]):
        acc, gyro, vdn, pdn = pathgen.calc_true_sensor_output(t('pos')[i], t('vel_b')[i], t('att')[i], t('c_nb')[i], t('vdot')[i],
                                                              t('adot')[i], rf, t('g')[i])
        # NumPy's 3x3 dot products may associate differently from the plain C sums: a few ulp of the terms
        np.testing.assert_allclose(acc, t('acc')[i], rtol=0, atol=4e-13)
        np.testing.assert_allclose(gyro, t('gyro')[i], rtol=0, atol=4e-15)
        np.testing.assert_allclose(vdn, t('vel_dot_n')[i], rtol=0, atol=4e-13)
        np.testing.assert_allclose(pdn, t('pos_dot_n')[i], rtol=1e-14, atol=1e-13 if rf == 1 else 1e-19)
    with pytest.raises(ValueError):
        pathgen.calc_true_sensor_output(np.zeros(2), np.zeros(3), np.zeros(3), np.eye(3), np.zeros(3), np.zeros(3), 0, 9.8)


def test_parse_motion_def_is_the_references():
    from gnss_ins_sim.pathgen import pathgen
    g = load_golden('leaves')
    for i in range(g['pm_seg'].shape[0]):
        a, v = pathgen.parse_motion_def(g['pm_seg'][i], g['pm_att'][i], g['pm_vel'][i])
        np.testing.assert_array_equal(a, g['pm_att_com'][i])
        np.testing.assert_array_equal(v, g['pm_vel_com'][i])
    with pytest.raises(ValueError, match='motion type'):
        pathgen.parse_motion_def([7, 0, 0, 0, 0, 0, 0, 1, 0], np.zeros(3), np.zeros(3))


def test_array_error_and_calc_data_err():
    from gnss_ins_sim.sim import ins_data_manager
    g = load_golden('leaves')
    mgr = ins_data_manager.InsDataMgr([100.0, 0.0, 0.0], 0)
    np.testing.assert_allclose(mgr.array_error(g['ae_ang_x'], g['ae_ang_r'], angle=True), g['ae_ang'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(mgr.array_error(g['ae_lla_x'], g['ae_lla_r'], lla=1), g['ae_ned'], rtol=0, atol=2e-9)
    np.testing.assert_allclose(mgr.array_error(g['ae_lla_x'], g['ae_lla_r'], lla=2), g['ae_ecef'], rtol=0, atol=2e-9)
    # calc_data_err: the Sim_data of the error series, per key, angles wrapped; the 'ned' option relabels a ref_frame-0 position
    mgr.add_data('time', np.arange(50) / 100.0)
    mgr.add_data('ref_att_euler', g['ae_ang_r'], units=['rad', 'rad', 'rad'])
    mgr.add_data('ref_pos', g['ae_lla_r'], units=['rad', 'rad', 'm'])
    mgr.set_algo_output(['att_euler', 'pos'])
    mgr.add_data('att_euler', {'algo0_0': g['ae_ang_x'], 'algo0_1': g['ae_ang_x'] + 0.25}, units=['rad', 'rad', 'rad'])
    mgr.add_data('pos', {'algo0_0': g['ae_lla_x']}, units=['rad', 'rad', 'm'])
    e = mgr.calc_data_err('att_euler', 'ref_att_euler', angle=True)
    assert e.name == 'err_att_euler' and e.description.startswith('ERROR of ') and sorted(e.data) == ['algo0_0', 'algo0_1']
    np.testing.assert_allclose(e.data['algo0_0'], g['ae_ang'], rtol=0, atol=1e-15)
    assert np.all(np.abs(e.data['algo0_1']) <= np.pi)
    p = mgr.calc_data_err('pos', 'ref_pos', err_opt='ned')
    assert p.units == ['m', 'm', 'm'] and p.legend == ['pos_N', 'pos_E', 'pos_D'] and p.description == 'ERROR of NED position'
    np.testing.assert_allclose(p.data['algo0_0'], g['ae_ned'], rtol=0, atol=2e-9)
    assert mgr.calc_data_err('vel', 'ref_vel') is None


@pytest.mark.gpu
def test_bias_drift_is_the_kernels_drift_term():
    """pathgen.bias_drift under its reference name = the drift the fused kernels add to the accelerometer of run 0 (the
    reference's function fed the same normals through the randn shim: golden bd_*)."""
    from gnss_ins_sim.pathgen import pathgen
    g = load_golden('leaves')
    bd = pathgen.bias_drift(g['bd_corr'], g['bd_drift'], int(g['bd_n']), float(g['bd_fs']), seed=int(g['bd_seed']))
    assert bd.shape == (int(g['bd_n']), 3)
    np.testing.assert_allclose(bd, g['bd_out'], rtol=0, atol=1e-17)


def test_names_off_the_path_say_where_they_are(monkeypatch):
    """Without a checkout: AttributeError / ImportError, and the message names the environment variable."""
    monkeypatch.delenv('GNSS_INS_SIM_REFERENCE', raising=False)
    code = ("import sys; sys.path[:0] = [%r]\n"
            "from gnss_ins_sim.attitude import attitude\n"
            "try:\n    attitude.quat_update\nexcept AttributeError as e:\n    assert 'GNSS_INS_SIM_REFERENCE' in str(e), e\nelse:\n    raise SystemExit('no error')\n"
            "try:\n    import demo_algorithms.inclinometer_mahony\nexcept ImportError:\n    pass\nelse:\n    raise SystemExit('imported')\n"
            "print('OK')\n" % PKG)
    env = {k: v for k, v in os.environ.items() if k != 'GNSS_INS_SIM_REFERENCE'}
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0 and b'OK' in out.stdout, out.stdout.decode()[-2000:]


_HOSTED = r'''
import os, sys, glob
import numpy as np
sys.path[:0] = [%(pkg)r]
before = set(glob.glob(%(ref)r + '/**/*.pyc', recursive=True))
from gnss_ins_sim.attitude import attitude
from gnss_ins_sim.geoparams import geoparams
from gnss_ins_sim.sim import ins_sim
import gnss_ins_sim, demo_algorithms
assert os.path.dirname(gnss_ins_sim.__file__).startswith(%(pkg)r) and os.path.dirname(demo_algorithms.__file__).startswith(%(pkg)r)

# names of the reference's modules that the drop-in does not define resolve to the reference's functions ...
q = attitude.quat_update(np.array([1.0, 0.0, 0.0, 0.0]), np.array([0.1, -0.2, 0.3]), 0.01)
assert abs(np.linalg.norm(q) - 1.0) < 1e-15 and attitude.quat_update.__module__.endswith('_reference_attitude')
c = attitude.quat2dcm(q)
assert np.allclose(attitude.dcm2quat(c), q)
xyz = geoparams.lla2ecef(np.array([0.5, 1.0, 100.0]))
assert np.allclose(geoparams.ecef2lla(xyz), [0.5, 1.0, 100.0], atol=1e-9)          # ecef2lla: reference; lla2ecef: ours
# ... while the hot-path leaves stay this package's
assert attitude.euler_update_zyx.__module__ == 'gnss_ins_sim.attitude.attitude'
# a module the drop-in lacks comes from the checkout under its usual name; ITS imports of hot-path modules get the drop-in's
import demo_algorithms.inclinometer_mahony as mahony
assert mahony.__file__.startswith(%(ref)r) and mahony.attitude is attitude
from gnss_ins_sim.psd import time_series_from_psd
assert time_series_from_psd.__file__.startswith(%(ref)r)
from gnss_ins_sim.geoparams import geomag
assert geomag.__file__.startswith(%(ref)r)


class Hosted(object):
    """A user plugin in the reference's style that calls attitude leaves step by step: Euler angles by euler_update_zyx (hot-path
    leaf, the drop-in's) and a quaternion by quat_update (off the path, the reference's)."""
    def __init__(self):
        self.input = ['fs', 'gyro']
        self.output = ['att_euler', 'att_quat']
        self.results = None
    def run(self, set_of_input):
        fs, gyro = set_of_input[0], set_of_input[1]
        dt = 1.0 / fs
        n = gyro.shape[0]
        eul, quat = np.zeros((n, 3)), np.zeros((n, 4))
        quat[0] = attitude.euler2quat(eul[0])
        for i in range(1, n):
            eul[i] = attitude.euler_update_zyx(eul[i - 1], gyro[i - 1], dt)
            quat[i] = attitude.quat_update(quat[i - 1], gyro[i - 1], dt)
        self.results = [eul, quat]
    def get_results(self):
        return self.results
    def reset(self):
        pass

# Sim over a directory of logged data (no GPU involved): the plugin is hosted the reference's way (ins_algo_manager.py:39-96)
d = sys.argv[1]
n, fs = 400, 100.0
t = np.arange(n) / fs
gyro = np.stack([0.2 * np.sin(t), 0.1 * np.cos(2 * t), 0.3 * np.ones(n)], 1)
np.savetxt(d + '/time.csv', t, header='time (sec)', comments='')
np.savetxt(d + '/gyro-0.csv', gyro * 180 / np.pi, delimiter=',', header='gyro_x (deg/s),gyro_y (deg/s),gyro_z (deg/s)', comments='')
sim = ins_sim.Sim([fs, 0.0, 0.0], d, ref_frame=0, imu=None, algorithm=Hosted())
sim.run(1)
eul = sim.dmgr.att_euler.data[0] if 0 in sim.dmgr.att_euler.data else list(sim.dmgr.att_euler.data.values())[0]
quat = sim.dmgr.att_quat.data[0] if 0 in sim.dmgr.att_quat.data else list(sim.dmgr.att_quat.data.values())[0]
# the two integrations describe the same rotation (first-order Euler-rate steps against exact quaternion steps)
d_ang = attitude.euler2quat(eul[-1])
assert abs(abs(np.dot(d_ang, quat[-1])) - 1.0) < 5e-4, (d_ang, quat[-1])
after = set(glob.glob(%(ref)r + '/**/*.pyc', recursive=True))
assert after == before, 'bytecode was written into the checkout: %%s' %% sorted(after - before)
print('OK')
'''


@need_ref
def test_hosted_plugin_reaches_reference_names_through_the_dropin(tmp_path):
    """With $GNSS_INS_SIM_REFERENCE naming a checkout, a hosted plugin that calls attitude.quat_update (attitude.py:665) and
    attitude.euler_update_zyx (:679) runs inside the drop-in Sim; demo_algorithms.inclinometer_mahony (which calls quat_update /
    dcm2quat at :115, 151) imports; nothing is written into the checkout."""
    script = tmp_path / 'hosted.py'
    script.write_text(_HOSTED % {'pkg': PKG, 'ref': REF})
    data = tmp_path / 'logged'
    data.mkdir()
    env = dict(os.environ, GNSS_INS_SIM_REFERENCE=REF, MPLBACKEND='Agg')
    env.pop('PYTHONDONTWRITEBYTECODE', None)             # the loader itself must keep the checkout clean
    out = subprocess.run([sys.executable, str(script), str(data)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and b'OK' in out.stdout, out.stdout.decode()[-4000:]
