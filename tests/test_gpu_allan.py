"""On-device Allan variance vs the reference's allan.allan_var (golden) and vs the oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import ginsim
    c = ginsim.Context(0)
    yield c
    c.close()


def _series(seed, n):
    from oracle import philox
    j = np.arange(n, dtype=np.uint64)
    return 0.3 * philox.normal_pair(seed, 7, 5, j)[0] + 1e-3 * np.cumsum(philox.normal_pair(seed, 7, 4, j)[1])


def test_allan_matches_reference_golden(ctx):
    import ginsim
    g = load_golden('allan_ref')
    x = _series(int(g['seed']), int(g['n']))
    avar, tau = ginsim.allan_var_host(ctx, x, float(g['fs']))
    np.testing.assert_allclose(tau, g['tau'], rtol=1e-15)
    np.testing.assert_allclose(avar, g['avar'], rtol=1e-10)          # SURVEY 8(c) T6 tolerance


@pytest.mark.parametrize('n,fs', [(360000, 100.0), (123457, 50.0), (2519, 10.0), (2521, 10.0), (25200, 100.0), (899, 100.0), (90, 10.0),
                                  (15120, 10.0), (15121, 10.0), (151200, 100.0), (151211, 100.0)])
def test_allan_matches_oracle_ragged_lengths(ctx, n, fs):
    import ginsim
    from oracle import ins_np
    x = np.stack([_series(s, n) + 5.0 * s for s in range(3)])          # large offsets: shift invariance
    avar, tau = ginsim.allan_var_host(ctx, x, fs)
    for s in range(3):
        ra, rt = ins_np.allan_var(x[s], fs)
        assert tau.shape == rt.shape
        np.testing.assert_allclose(tau, rt, rtol=1e-15)
        np.testing.assert_allclose(avar[s], ra, rtol=1e-9)


@pytest.mark.parametrize('n', [2520 * 4, 2520 * 4 + 9, 2520 * 40 - 1, 25200 * 3 + 2527, 252000 + 10])
def test_allan_chunk_boundaries_strided_series_and_drift(ctx, n):
    """Lengths around multiples of the 2520-entry chunk (its 9-entry halo, the hand-over between the per-chunk levels
    and the single-chunk tail levels), series packed with a stride larger than n, and a bias + ramp a million times
    the noise: the per-chunk origin shift must keep the bin-sum differences exact."""
    import ginsim
    from oracle import ins_np
    S, stride, fs = 4, n + 13, 100.0
    t = np.arange(n) / fs
    rows = [_series(10 + s, n) + 1.0e6 * (s + 1) + 3.0e3 * s * t for s in range(S)]
    packed = np.full((S, stride), np.nan)
    for s in range(S):
        packed[s, :n] = rows[s]
    buf = ctx.upload(packed)
    avar, tau = ginsim.allan_var(ctx, buf, n, S, stride, fs)
    for s in range(S):
        ra, rt = ins_np.allan_var(rows[s], fs)
        assert tau.shape == rt.shape and np.isfinite(avar[s]).all()
        np.testing.assert_allclose(tau, rt, rtol=1e-15)
        # the reference's own means of ~1e6-sized samples carry 1e-10 of rounding relative to a noise-sized difference
        np.testing.assert_allclose(avar[s], ra, rtol=2e-7)


@pytest.mark.parametrize('n', [25200 * 3, 25200 * 3 + 2, 25200 * 3 - 2, 25200 * 2 + 2520 * 7 + 8, 25220, 25200 * 6 + 90, 25200 * 4 + 25198,
                               252000 * 2 + 2520 * 3 + 1234])
def test_allan_fused_levels_at_their_chunk_boundaries(ctx, n, monkeypatch):
    """Round 5: levels 0 and 1 in one launch -- a workgroup takes ten chunks of level 0 = one 2520-entry chunk of level 1 and
    the pairs of level-1 bins across workgroups are added by the finishing launch.  Lengths around multiples of 25 200 (the
    series' last workgroup ragged, with fewer than ten chunks, or with level-1 bins that do not exist), a level 1 of barely more
    than one chunk, 16-byte aligned rows with a stride larger than n, bias + ramp a million times the noise; against the
    oracle, and against the two-launch form of the same library (GINSIM_ALLAN_FUSE=0)."""
    import ginsim
    from oracle import ins_np
    S, stride, fs = 3, n + 14, 100.0
    t = np.arange(n) / fs
    rows = [_series(20 + s, n) + 1.0e6 * (s + 1) + 3.0e3 * s * t for s in range(S)]
    packed = np.full((S, stride), np.nan)
    for s in range(S):
        packed[s, :n] = rows[s]
    buf = ctx.upload(packed)
    monkeypatch.setenv('GINSIM_ALLAN_FUSE', '1')
    avar, tau = ginsim.allan_var(ctx, buf, n, S, stride, fs)
    monkeypatch.setenv('GINSIM_ALLAN_FUSE', '0')
    avar2, tau2 = ginsim.allan_var(ctx, buf, n, S, stride, fs)
    np.testing.assert_array_equal(tau, tau2)

    # level 0 is the same arithmetic; level 1 differs in how the cross-workgroup pairs are formed, later levels in nothing
    np.testing.assert_allclose(avar, avar2, rtol=1e-11)
    assert not np.array_equal(avar, avar2) or n < 25200 * 2        # the fused form really ran (its sums associate differently)
    for s in range(S):
        ra, rt = ins_np.allan_var(rows[s], fs)
        assert tau.shape == rt.shape and np.isfinite(avar[s]).all()
        np.testing.assert_allclose(tau, rt, rtol=1e-15)
        np.testing.assert_allclose(avar[s], ra, rtol=2e-7)
    buf.free()


def test_allan_plugin_and_module_surface(ctx):
    from gnss_ins_sim.allan import allan
    from demo_algorithms import allan_analysis
    from oracle import ins_np
    n, fs = 50000, 100.0
    acc = np.stack([_series(s, n) for s in range(3)], axis=1)
    gyr = np.stack([_series(s + 3, n) * 0.01 for s in range(3)], axis=1)
    a = allan_analysis.Allan()
    a.run([fs, acc, gyr])
    tau, ad_a, ad_g = a.get_results()
    ra, rt = ins_np.allan_var(gyr[:, 1], fs)
    np.testing.assert_allclose(tau, rt)
    np.testing.assert_allclose(ad_g[:, 1], np.sqrt(ra), rtol=1e-9)
    assert ad_a.shape == (tau.size, 3)
    av, t2 = allan.allan_var(acc[:, 0], fs)
    np.testing.assert_allclose(av, ins_np.allan_var(acc[:, 0], fs)[0], rtol=1e-9)
    assert allan.allan_var(acc[:50, 0], fs) == ([], [])


def test_allan_full_size_white_noise_slope(ctx):
    """Config 5 size: 3600 s @ 400 Hz (n = 1 440 000, 46 tau).  White noise of density N: AD(tau) = N/sqrt(tau)."""
    import ginsim
    n, fs, N = 1440000, 400.0, 7.27e-5
    z0, _ = ginsim.rng_normals(ctx, 11, 0, 5, n)
    x = N * np.sqrt(fs) * z0
    avar, tau = ginsim.allan_var_host(ctx, x, fs)
    assert tau.size == 46 and tau[-1] == 250.0
    ad = np.sqrt(avar)
    # sampling error of an Allan deviation from nb bins: ~ 1 / sqrt(2 (nb - 1)) relative (1 sigma); held to 4 sigma + 0.5 %
    nb = np.floor(n / (tau * fs))
    tol = 4.0 / np.sqrt(2.0 * (nb - 1.0)) + 0.005
    k = tau <= 10.0
    assert np.all(np.abs(ad[k] / (N / np.sqrt(tau[k])) - 1.0) < tol[k]), (ad[k] / (N / np.sqrt(tau[k])) - 1.0, tol[k])


def test_allan_config5_size_against_the_oracle(ctx):
    """BASELINE config 5 at its real size: 3600 s static @ 400 Hz (n = 1 440 000), 32 runs -> 192 series generated on the device
    by the time-parallel sensor kernels (mid-accuracy IMU: white noise + Gauss-Markov drift), re-laid out and analysed by ONE
    Allan call with the 192-series batch grid (46 averaging factors; the levels 1 440 000 / 144 000 / 14 400 run chunked, 1440 /
    144 / 14 in the finishing launch).  Twelve series spread over the batch (first / last run, every axis of both sensors) are
    pulled to the host and pushed through the oracle (allan.py:18-59 restated): all 46 tau to 1e-10."""
    import ginsim
    from ginsim import workloads
    from oracle import ins_np, c_oracle
    fs, seconds, runs = 400.0, 3600.0, 32
    text = open(workloads.profile_path('static_1800s')).read().split('\n')
    ini, _ = workloads.parse_motion('\n'.join(text[:4]))
    raw = ginsim.pathgen(ini, np.array([[1.0, 0, 0, 0, 0, 0, 0, seconds, 0.0]]), fs, 0.0, workloads.HIGH_MOBILITY, 1)
    truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
             'ref_pos': raw['nav'][:, 1:4], 'ref_vel': raw['nav'][:, 4:7], 'ref_att': raw['nav'][:, 7:10]}
    n = truth['ref_accel'].shape[0]
    assert n == 1440000
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ctx, fs, 1, truth, acc, gyr, None, runs=runs, algos=(), seed=20260924, keep_sensors=True).run()
    tau, ad = job.allan(fs)
    assert tau.size == 46 and tau[0] == 1.0 / fs and tau[-1] == 250.0
    assert ad['gyro'].shape == (runs, 46, 3)
    worst = 0.0
    for name in ('accel', 'gyro'):
        series = job.sensors(name, [0, runs - 1])                  # (2, n, 3) host copies
        for k, r in enumerate((0, runs - 1)):
            for ax in range(3):
                # the NumPy oracle: reshape + mean as allan.py:44-50 does it (pairwise summation).  The plain-C restatement sums
                # a bin sequentially and is the LESS accurate side on the accel-z series (offset -9.79 m/s^2 under 5e-3 of noise:
                # 1e-9 at tau = 250 s against a long-double evaluation, where the device and NumPy agree to 1e-11)
                va, ta = ins_np.allan_var(series[k][:, ax], fs)
                np.testing.assert_allclose(tau, ta, rtol=1e-15)
                np.testing.assert_allclose(ad[name][r][:, ax], np.sqrt(va), rtol=1e-10)
                worst = max(worst, float(np.abs(ad[name][r][:, ax] / np.sqrt(va) - 1.0).max()))
                if name == 'gyro' or ax < 2:                        # zero-mean series: the C oracle as well
                    vc, _ = c_oracle.allan_var(series[k][:, ax], fs)
                    np.testing.assert_allclose(ad[name][r][:, ax], np.sqrt(vc), rtol=1e-10)
    try:
        from test_gpu_full_size import _record
        _record('c5_allan_192x1440000_vs_oracle', rel=worst)
    except ImportError:
        pass
    # the model shows: white noise -1/2 slope at short tau (ARW), the Gauss-Markov bump above it at long tau
    k1 = int(np.argmin(np.abs(tau - 1.0)))
    assert abs(ad['gyro'][:, k1, 0].mean() / float(np.asarray(gyr['arw'])[0]) - 1.0) < 0.02
    job.release()


@pytest.mark.parametrize('runs', [1, 3])
def test_sim_allan_flow_stays_on_the_device_and_matches_the_host_plugin(ctx, runs):
    """demo_allan.py's flow (BASELINE config 5, second half): Sim on a static profile with the Allan plugin.  Inside this
    package's Sim the plugin gets the device-resident sensor series of all runs (run_device: time-parallel generation for
    few runs, [3][n][R] -> [R][3][n] re-layout, one Allan call per sensor); the result per run must equal the plugin's
    plain run() on the host copy of that run's series, and the model's ARW must show up as the -1/2 slope."""
    import os
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import allan_analysis
    from conftest import PKG
    csv = os.path.join(PKG, 'motion_profiles', 'static_1800s.csv')
    with open(csv) as f:
        lines = f.read().splitlines()
    short = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'static_short_%d.csv' % os.getpid())
    cmd = lines[3].split(',')
    cmd[7] = '120'                                   # 120 s instead of 1800 s
    with open(short, 'w') as f:
        f.write('\n'.join(lines[:3] + [','.join(cmd)]) + '\n')
    fs = 100.0
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    algo = allan_analysis.Allan()
    sim = ins_sim.Sim([fs, 0.0, 0.0], short, ref_frame=1, imu=imu, mode=None, env=None, algorithm=algo, seed=5)
    sim.run(runs)
    tau = sim.dmgr.algo_time.data
    ad_g, ad_a = sim.dmgr.ad_gyro.data, sim.dmgr.ad_accel.data
    assert len(ad_g) == runs and len(ad_a) == runs
    from oracle import ins_np
    host = allan_analysis.Allan()
    for r in range(runs):
        key = 'algo0_%d' % r
        acc_r, gyr_r = sim.dmgr.accel.data[r], sim.dmgr.gyro.data[r]
        # the ORACLE (allan.py:18-59 restated in NumPy) on the host copy of this run's series, axis by axis
        for ax in range(3):
            va, ta = ins_np.allan_var(gyr_r[:, ax], fs)
            np.testing.assert_allclose(tau[key], ta, rtol=1e-15)
            np.testing.assert_allclose(ad_g[key][:, ax], np.sqrt(va), rtol=1e-9)
            va, _ = ins_np.allan_var(acc_r[:, ax], fs)
            np.testing.assert_allclose(ad_a[key][:, ax], np.sqrt(va), rtol=1e-9)
        # and the plugin's plain run(set_of_input) on host arrays (the form the reference's own Sim calls)
        host.run([fs, acc_r, gyr_r])
        t_h, a_h, g_h = host.get_results()
        np.testing.assert_allclose(tau[key], t_h, rtol=1e-15)
        np.testing.assert_allclose(ad_g[key], g_h, rtol=1e-9)
        np.testing.assert_allclose(ad_a[key], a_h, rtol=1e-9)
    # ARW 0.25 deg/sqrt(hr) = 7.27e-5 rad/s/sqrt(Hz): AD(tau = 1 s) within the estimator's scatter (12 000 samples)
    k = int(np.argmin(np.abs(tau['algo0_0'] - 1.0)))
    assert abs(ad_g['algo0_0'][k, 0] / (0.25 / 60 * np.pi / 180) - 1.0) < 0.25
    os.remove(short)


@pytest.mark.parametrize('C,n,R', [(3, 1000, 32), (3, 1, 1), (1, 63, 5), (2, 64, 64), (3, 65, 70), (1, 130, 130), (3, 4097, 33), (2, 129, 1)])
def test_runs_to_series_relayout_every_tile_shape(ctx, C, n, R):
    """[C][n][R] -> [R][C][n] on the device (what the Allan flow does with the kept sensor series of many runs): full and ragged
    tiles in both directions, runs that are and are not a power of two, fewer runs than a wavefront has lanes."""
    import ginsim
    from ginsim._lib import check
    rng = np.random.default_rng(C * 1000003 + n * 131 + R)
    x = rng.normal(size=(C, n, R))
    src = ctx.upload(x)
    dst = ctx.malloc(8 * x.size)
    check(ginsim.lib.ginsim_runs_to_series(ctx.handle, src.ptr, C, n, R, dst.ptr))
    got = ctx.download(dst, (R, C, n))
    assert np.array_equal(got, np.ascontiguousarray(x.transpose(2, 0, 1)))
    src.free()
    dst.free()
