"""Pin the plain-C oracle against the reference-generated goldens and against the NumPy oracle. CPU only."""
import numpy as np
import pytest

from oracle import c_oracle, ins_np, philox
from conftest import load_golden, assert_traj_close, ang_close, golden_vibration, T3_VIB


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


@pytest.mark.parametrize('name', ['t3_demo_rf1', 't3_mid_rf0', 't3_white_gps_rf0', 't3_low_rf1', 't3_high_odo_rf0', 't3_drive200_rf0'] + T3_VIB)
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
        vib_acc, vib_gyro = golden_vibration(g)
        end, traj, sens = c_oracle.mc_run(int(g['seed']), 0, R, fs, rf, truth, acc_err, gyr_err, g['ini'], algo=a,
                                          odo_err=odo_err, keep=R, vib_accel=vib_acc, vib_gyro=vib_gyro)
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


# ------------------------------------------------------------------ the FLOAT restatement (fp32 kernel path, BASELINE config 5)
# Stated fp32 tolerances against the reference's fp64 outputs (10 s / 1000 steps unless noted): attitude 2e-6 rad (the float
# ulp of a 5.5 rad yaw is 4.8e-7), velocity 5e-5 m/s, position 1e-4 m (ref_frame 1: ECEF via the fp64 displacement;
# ref_frame 0: lat / lon 1e-11 rad, altitude 1e-4 m); sensors: accel 2e-6 m/s^2 (float ulp at 9.8 is 9.5e-7), gyro 1e-7 rad/s.
F32_TOL = {'att': 2e-6, 'vel': 5e-5, 'pos_m': 1e-4, 'pos_rad': 1e-11, 'accel': 2e-6, 'gyro': 1e-7}


def assert_f32_close(att, dpos, vel, ini, rf, g_att, g_pos, g_vel, what='', scale=1.0, att_tol=None):
    from gnss_ins_sim.geoparams import geoparams
    pos0 = geoparams.lla2ecef(np.asarray(ini[:3], dtype=np.float64)) if rf == 1 else np.asarray(ini[:3], dtype=np.float64)
    pos = dpos.astype(np.float64) + pos0
    assert ang_close(att, g_att, att_tol or F32_TOL['att']), what + ' att'
    assert np.abs(vel - g_vel).max() <= F32_TOL['vel'] * scale, what + ' vel %.2e' % np.abs(vel - g_vel).max()
    dp = np.abs(pos - g_pos)
    if rf == 1:
        assert dp.max() <= F32_TOL['pos_m'] * scale, what + ' pos %.2e' % dp.max()
    else:
        assert dp[..., :2].max() <= F32_TOL['pos_rad'] * scale and dp[..., 2].max() <= F32_TOL['pos_m'] * scale, what + ' pos'


@pytest.mark.parametrize('name', ['bosch', 'nxp', 'tumble'])
def test_c_f32_t1_fixture(name):
    """oracle_free_integration_f32 on the reference's logged-IMU fixtures (and the tumble that drives the pitch over the pole
    twice: cos(pitch) down to 6e-3 amplifies float rounding, stated attitude tolerance 1e-4 rad there)."""
    g = load_golden('t1_fixture_' + name)
    k = g['rows']
    for tag, rf, ini, erot in (('extg', 0, g['ini'], False), ('wgs', 0, g['ini'][:9], True), ('rf1', 1, g['ini'][:9], True)):
        att, dpos, vel, _ = c_oracle.free_integration_f32(rf, float(g['fs']), g['gyro'], g['accel'], ini, earth_rot=erot)
        assert_f32_close(att[k], dpos[k], vel[k], ini, rf, g['att_' + tag], g['pos_' + tag], g['vel_' + tag], what=name + tag,
                         att_tol=1e-4 if name == 'tumble' else None)


@pytest.mark.parametrize('name', ['t3_demo_rf1', 't3_mid_rf0', 't3_white_gps_rf0', 't3_low_rf1', 't3_high_odo_rf0', 't3_drive200_rf0'])
def test_c_f32_t3_injected_noise(name):
    """The float restatement consumes the SAME normals as the fp64 path: against the reference executed with those normals
    injected, every sensor sample and every trajectory sample of every run agrees within the stated fp32 tolerances
    (t3_drive200: 4400 steps / 22 s, position and velocity tolerances x 5)."""
    g = load_golden(name)
    R, k, fs, rf = int(g['R']), g['rows'], float(g['fs']), int(g['ref_frame'])
    acc_err, gyr_err = _errs(g)
    truth = {'ref_accel': g['ref_accel'], 'ref_gyro': g['ref_gyro'], 'ref_att': g['ref_att'], 'ref_pos': g['ref_pos'],
             'ref_vel': g['ref_vel']}
    odo_err = None
    if 'odo' in g:
        truth['ref_odo'] = g['ref_odo']
        odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])}
    scale = 5.0 if g['ref_accel'].shape[0] > 1000 else 1.0
    for a, tag in (('free', 'fi'), ('odo', 'odo')):
        if tag + '_att' not in g:
            continue
        end, traj, sens, odo = c_oracle.mc_run_f32(int(g['seed']), 0, R, fs, rf, truth, acc_err, gyr_err, g['ini'], algo=a,
                                                   odo_err=odo_err, keep=R)
        np.testing.assert_allclose(sens[:, k, 0:3], g['accel'], rtol=0, atol=F32_TOL['accel'])
        np.testing.assert_allclose(sens[:, k, 3:6], g['gyro'], rtol=0, atol=F32_TOL['gyro'])
        if odo is not None:
            np.testing.assert_allclose(odo[:, k], g['odo'], rtol=0, atol=2e-6)
        assert_f32_close(traj[:, k, 0:3], traj[:, k, 3:6], traj[:, k, 6:9], g['ini'], rf, g[tag + '_att'], g[tag + '_pos'],
                         g[tag + '_vel'], what=name + a, scale=scale)
        # and the end-point error record against the fp64 restatement of the same runs
        end64, _, _ = c_oracle.mc_run(int(g['seed']), 0, R, fs, rf, truth, acc_err, gyr_err, g['ini'], algo=a, odo_err=odo_err)
        assert ang_close(end[:, :3], end64[:, :3], F32_TOL['att'])
        np.testing.assert_allclose(end[:, 6:9], end64[:, 6:9], rtol=0, atol=F32_TOL['vel'] * scale)


def test_c_f32_noise_free_closed_loop():
    """T2 in float: truth through the float sensor model with zero noise -> mechanisation -> the reference's closed-loop rows."""
    import ginsim
    from ginsim import workloads
    zero = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, 100.0), 'arw': np.zeros(3), 'vrw': np.zeros(3)}
    for rf in (0, 1):
        g = load_golden('t2_turn_rf%d' % rf)
        ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
        k = g['rows']
        for a, tag in (('free', 'fi'), ('odo', 'odo')):
            _, traj, sens, _ = c_oracle.mc_run_f32(1, 0, 1, 100.0, rf, truth, zero, zero, ini, algo=a, odo_err={'scale': 1.0, 'stdv': 0.0},
                                                   keep=1)
            assert_f32_close(traj[0, k, 0:3], traj[0, k, 3:6], traj[0, k, 6:9], ini, rf, g[tag + '_att'], g[tag + '_pos'],
                             g[tag + '_vel'], what='t2 rf%d %s' % (rf, a))
            assert np.array_equal(sens[0, :, 3:6], truth['ref_gyro'].astype(np.float32))


def test_c_f32_sincos_def_is_float_accurate():
    """The defined sin / cos (fp64 reduction + Taylor, rounded to float once) is the correctly rounded float of the true value
    on the angles the mechanisation uses (|x| <= 2 pi), except where fp64 libm itself would round the other way (none here)."""
    import ctypes as C
    lib = c_oracle.lib()
    rng = np.random.RandomState(4)
    x = np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, 200000), np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, 5.497787143782138])])
    att, dpos, vel = (np.empty((2, 3), dtype=np.float32) for _ in range(3))
    bad = 0
    for chunk in np.array_split(x, 50):
        for v in chunk[:200]:            # through the public entry: attitude (v, 0, 0) -> att cached trig drives vel = C^T [1, 0, 0]
            ini = np.array([0.5, 1.0, 0.0, 1.0, 0.0, 0.0, v, 0.0, 0.0])
            a, _, vv, _ = c_oracle.free_integration_f32(1, 100.0, np.zeros((2, 3)), np.zeros((2, 3)), ini)
            vf = np.float64(np.float32(v))
            want = np.array([np.float32(np.cos(vf)), np.float32(np.sin(vf))])
            bad += int(abs(float(vv[0, 0]) - float(want[0])) > 6e-8) + int(abs(float(vv[0, 1]) - float(want[1])) > 6e-8)
    assert bad == 0


def test_c_f32_both_plugins_at_other_rates():
    """The float restatement of BOTH plugins' given-data form at 50 / 200 / 400 Hz (1600 samples: 32 / 8 / 4 s) against the
    reference-executed golden: attitude 2e-6 rad, velocity and position tolerances scaled with the duration (x 3.2 at 32 s)."""
    from test_oracle_golden import _t1_rates_cases
    worst = {}
    for c, tag, rf, ini, erot, plug in _t1_rates_cases():
        k = c['rows']
        fs = float(c['fs'])
        att, dpos, vel, _ = c_oracle.free_integration_f32(rf, fs, c['gyro'], c['accel'] if plug == 'free' else None, ini, earth_rot=erot,
                                                          odo=c['odo'] if plug == 'odo' else None)
        scale = max(1.0, 1600.0 / fs / 10.0) ** 2
        assert_f32_close(att[k], dpos[k], vel[k], ini, rf, c['%s_%s_att' % (plug, tag)], c['%s_%s_pos' % (plug, tag)],
                         c['%s_%s_vel' % (plug, tag)], what='%s %s %g Hz' % (plug, tag, fs), scale=scale, att_tol=4e-6)
