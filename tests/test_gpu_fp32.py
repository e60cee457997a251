"""fp32 kernel path (BASELINE config 5) against fp64 / the reference.  STATED TOLERANCES for the 90-degree turn
(n = 1000 steps, 10 s):
  * noise-free closed loop vs the reference's fp64 outputs: attitude 2e-6 rad, velocity 5e-5 m/s,
    position 1e-4 m (ref_frame 1: ECEF+displacement) / 1e-11 rad + 1e-4 m (ref_frame 0: lat, lon, alt);
    measured: 2.2e-7 rad, 7.6e-6 m/s, 1.4e-5 m, 3e-12 rad;
  * with noise the fp32 path uses 23-bit uniforms (a different, coarser stream than fp64), so the comparison is
    statistical: end-point std of 65 536 runs within 1.5 % of the fp64 path (sampling error of the ratio 0.4 %),
    means within 5 sigma/sqrt(R); generated white noise has the model's sigma within 1 %.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
ZERO = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, 100.0), 'arw': np.zeros(3), 'vrw': np.zeros(3)}


@pytest.fixture(scope='module')
def ctx():
    import ginsim
    c = ginsim.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize('rf', [0, 1])
def test_fp32_noise_free_vs_reference(ctx, rf):
    import ginsim
    from ginsim import workloads
    g = load_golden('t2_turn_rf%d' % rf)
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, ZERO, ZERO, ini, runs=130, algos=('free', 'odo'),
                               odo_err={'scale': 1.0, 'stdv': 0.0}, seed=1, keep_sensors=True, keep_traj=True,
                               precision='f32').run()
    k = g['rows']
    for a, tag in (('free', 'fi'), ('odo', 'odo')):
        att, pos, vel = job.trajectories(a, [0, 129])
        for r in range(2):
            d = np.mod(att[r][k] - g[tag + '_att'] + np.pi, 2 * np.pi) - np.pi
            assert np.abs(d).max() < 2e-6
            assert np.abs(vel[r][k] - g[tag + '_vel']).max() < 5e-5
            dp = np.abs(pos[r][k] - g[tag + '_pos'])
            if rf == 1:
                assert dp.max() < 1e-4
            else:
                assert dp[:, :2].max() < 1e-11 and dp[:, 2].max() < 1e-4
    np.testing.assert_allclose(job.sensors('gyro', [5])[0], truth['ref_gyro'], rtol=0, atol=1e-7)
    e = job.end_errors('free')
    assert np.all(e[0] == e[129])            # deterministic and lane independent
    job.release()


@pytest.mark.parametrize('rf', [0, 1])
def test_fp32_statistics_match_fp64(ctx, rf):
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    R = 65536
    st = {}
    for prec in ('f64', 'f32'):
        job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, seed=5, precision=prec).run()
        st[prec] = job.stats('free')
        job.release()
    np.testing.assert_allclose(st['f32'].std, st['f64'].std, rtol=0.015)
    se = np.sqrt(st['f64'].std ** 2 + st['f32'].std ** 2) / np.sqrt(R)
    assert np.all(np.abs(st['f32'].mean - st['f64'].mean) < 5 * se + 2e-6 * np.abs(st['f64'].mean))


def test_fp32_generated_noise_moments(ctx):
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    odo_err = {'scale': 0.999, 'stdv': 0.1}
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=256, algos=('free', 'odo'), odo_err=odo_err, seed=3,
                               keep_sensors=True, precision='f32').run()
    runs = np.arange(0, 256, 4)
    for name, ref, sig in (('accel', truth['ref_accel'], acc['vrw'] * 10.0), ('gyro', truth['ref_gyro'], gyr['arw'] * 10.0)):
        e = job.sensors(name, runs) - ref[None]
        d = np.diff(e, axis=1) / np.sqrt(2.0)            # differencing removes the slow Gauss-Markov drift
        np.testing.assert_allclose(d.std(axis=(0, 1)), sig, rtol=0.01)
        z = d / d.std(axis=(0, 1))
        assert np.all(np.abs((z ** 4).mean(axis=(0, 1)) - 3.0) < 0.08) and np.all(np.abs(z.mean(axis=(0, 1))) < 0.02)
    eo = job.sensors('odo', runs) - 0.999 * truth['ref_odo'][None]
    np.testing.assert_allclose(eo.std(), 0.1, rtol=0.01)
    job.release()


def test_fp32_normals_cut_from_two_blocks_are_standard_and_independent(ctx):
    """The twelve fp32 normals of a step come from two Philox blocks (23-bit radius + 18-bit angle per Box-Muller pair,
    some angle bits being the spare low bits of radius words).  With zero drift the kept sensor series minus truth are
    sigma * N per axis: the six axes must be standard normal (Kolmogorov-Smirnov), mutually uncorrelated -- also in
    their squares, which would expose shared radius bits -- and white along time."""
    import ginsim
    from ginsim import workloads
    from scipy import stats
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, 100.0), 'vrw': np.full(3, 0.03 / 60.0)}
    gyr = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, 100.0), 'arw': np.full(3, 0.25 * np.pi / 180 / 60.0)}
    R = 256
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=123, keep_sensors=True, precision='f32').run()
    runs = np.arange(R)
    za = (job.sensors('accel', runs) - truth['ref_accel'][None]) / (acc['vrw'] * 10.0)      # white = rw / sqrt(dt), dt = 0.01
    zg = (job.sensors('gyro', runs) - truth['ref_gyro'][None]) / (gyr['arw'] * 10.0)
    z = np.concatenate([za, zg], axis=2)[:, :-1, :]                   # (R, n-1, 6)
    flat = z.reshape(-1, 6).T
    nsmp = flat.shape[1]
    lim = 5.0 / np.sqrt(nsmp)
    # accel z rides on -9.8 m/s^2 in fp32: its quantisation (ulp 9.5e-7 against sigma 5e-3) is far below the limits used
    for k in range(6):
        assert stats.kstest(flat[k][::7], 'norm').pvalue > 1e-4, 'axis %d fails KS' % k
        assert abs(flat[k].mean()) < lim and abs(flat[k].var() - 1.0) < 5.0 * np.sqrt(2.0 / nsmp) + 2e-3
        assert abs(stats.kurtosis(flat[k])) < 5.0 * np.sqrt(24.0 / nsmp) + 1e-2
    c, c2 = np.corrcoef(flat), np.corrcoef(flat ** 2)
    for a in range(6):
        for b in range(a + 1, 6):
            assert abs(c[a, b]) < lim and abs(c2[a, b]) < lim + 2e-3, (a, b, c[a, b], c2[a, b])
        lag = np.mean(z[:, :-1, a] * z[:, 1:, a])
        assert abs(lag) < lim, (a, lag)
    job.release()
