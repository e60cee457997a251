"""Pin the plain-C oracle against the reference-generated goldens and against the NumPy oracle. CPU only."""
import numpy as np
import pytest

from oracle import c_oracle, ins_np, philox
from conftest import load_golden, assert_traj_close, ang_close


def test_c_normals_match_numpy():
    for seed, run, stream in ((0, 0, 0), (20260923, 3, 5), (2 ** 63 + 5, 2 ** 40 + 7, 17), (5, 6, 7)):
        z0, z1 = c_oracle.normals(seed, run, stream, 200000)
        r0, r1 = philox.normal_pair(seed, run, stream, np.arange(200000, dtype=np.uint64))
        assert np.array_equal(z0.view(np.uint64), r0.view(np.uint64))          # the transform is defined to the bit
        assert np.array_equal(z1.view(np.uint64), r1.view(np.uint64))


@pytest.mark.parametrize('name', ['bosch', 'nxp', 'tumble'])
def test_c_t1_fixture(name):
    g = load_golden('t1_fixture_' + name)
    k = g['rows']
    for tag, rf, ini, erot in (('extg', 0, g['ini'], False), ('wgs', 0, g['ini'][:9], True), ('rf1', 1, g['ini'][:9], True)):
        att, pos, vel = c_oracle.free_integration(rf, float(g['fs']), g['gyro'], g['accel'], ini, earth_rot=erot)
        assert_traj_close(att[k], pos[k], vel[k], g['att_' + tag], g['pos_' + tag], g['vel_' + tag], rtol=1e-11)


def _errs(g):
    acc = {k[6:]: g[k] for k in g if k.startswith('accel_') and k != 'accel'}
    gyr = {k[5:]: g[k] for k in g if k.startswith('gyro_') and k != 'gyro'}
    return acc, gyr


@pytest.mark.parametrize('name', ['t3_demo_rf1', 't3_mid_rf0', 't3_white_gps_rf0', 't3_low_rf1', 't3_high_odo_rf0', 't3_drive200_rf0'])
def test_c_t3_injected_noise(name):
    g = load_golden(name)
    R, k, fs, rf = int(g['R']), g['rows'], float(g['fs']), int(g['ref_frame'])
    acc_err, gyr_err = _errs(g)
    truth = {'ref_accel': g['ref_accel'], 'ref_gyro': g['ref_gyro'], 'ref_att': g['ref_att'], 'ref_pos': g['ref_pos'],
             'ref_vel': g['ref_vel']}
    odo_err = None
    if 'odo' in g:
        truth['ref_odo'] = g['ref_odo']
        odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])}
    for a, tag in (('free', 'fi'), ('odo', 'odo')):
        if tag + '_att' not in g:
            continue
        end, traj, sens = c_oracle.mc_run(int(g['seed']), 0, R, fs, rf, truth, acc_err, gyr_err, g['ini'], algo=a,
                                          odo_err=odo_err, keep=R)
        np.testing.assert_allclose(sens[:, k, 0:3], g['accel'], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sens[:, k, 3:6], g['gyro'], rtol=0, atol=1e-14)
        assert_traj_close(traj[:, k, 0:3], traj[:, k, 3:6], traj[:, k, 6:9], g[tag + '_att'], g[tag + '_pos'],
                          g[tag + '_vel'], rtol=1e-10, what=name + a)
        e = ins_np.end_point_errors(traj[:, :, 0:3], traj[:, :, 3:6], traj[:, :, 6:9], g['ref_att'], g['ref_pos'], g['ref_vel'])
        assert ang_close(end[:, :3], e[:, :3], 1e-12)
        np.testing.assert_allclose(end[:, 3:], e[:, 3:], rtol=0, atol=1e-9)


def test_c_long_drive_noise_free_end_state():
    """n = 193 036 samples, ref_frame 0: native pathgen (C ABI, host code) reproduces the reference truth rows,
    and the C oracle integrating that truth reproduces the reference's noise-free FreeIntegration rows."""
    import ginsim
    g = load_golden('t2_long_drive_rf0')
    r = ginsim.pathgen(g['ini_pva'], g['motion_def'], 200.0, 10.0, g['mobility'], 0, gps=True)
    assert r['imu'].shape[0] == int(g['n']) and r['gps'].shape[0] == int(g['m'])
    k = g['rows']
    np.testing.assert_allclose(r['imu'][k], g['imu'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['nav'][k], g['nav'], rtol=1e-15, atol=1e-13)
    np.testing.assert_allclose(r['gps'][g['gps_rows']], g['gps'], rtol=1e-15, atol=1e-13)
    np.testing.assert_allclose(r['odo'][k], g['odo'], rtol=1e-15, atol=1e-13)
    att, pos, vel = c_oracle.free_integration(0, 200.0, r['imu'][:, 4:7], r['imu'][:, 1:4], g['ini_pva'])
    assert ang_close(att[k], g['fi_att'], 1e-10)
    np.testing.assert_allclose(pos[k, :2], g['fi_pos'][:, :2], rtol=0, atol=1e-12)     # lat/lon [rad]
    np.testing.assert_allclose(pos[k, 2], g['fi_pos'][:, 2], rtol=1e-7, atol=1e-7)     # altitude [m]
    np.testing.assert_allclose(vel[k], g['fi_vel'], rtol=1e-7, atol=1e-9)


def test_c_allan():
    g = load_golden('allan_ref')
    n, fs, seed = int(g['n']), float(g['fs']), int(g['seed'])
    j = np.arange(n, dtype=np.uint64)
    x = 0.3 * philox.normal_pair(seed, 7, 5, j)[0] + 1e-3 * np.cumsum(philox.normal_pair(seed, 7, 4, j)[1])
    avar, tau = c_oracle.allan_var(x, fs)
    np.testing.assert_allclose(tau, g['tau'], rtol=1e-15)
    np.testing.assert_allclose(avar, g['avar'], rtol=1e-10)


def test_c_t1_both_plugins_at_other_rates():
    from test_oracle_golden import _t1_rates_cases
    for c, tag, rf, ini, erot, plug in _t1_rates_cases():
        k = c['rows']
        att, pos, vel = c_oracle.free_integration(rf, float(c['fs']), c['gyro'], c['accel'] if plug == 'free' else None, ini,
                                                  earth_rot=erot, odo=c['odo'] if plug == 'odo' else None)
        assert_traj_close(att[k], pos[k], vel[k], c['%s_%s_att' % (plug, tag)], c['%s_%s_pos' % (plug, tag)],
                          c['%s_%s_vel' % (plug, tag)], rtol=1e-11, what='%s %s %g Hz' % (plug, tag, float(c['fs'])))
