"""fp32 kernel path (BASELINE config 5).  Three levels of evidence:

  1. BIT-EXACT against the float restatement in oracle/c/ginsim_oracle.c (oracle_mc_run_f32): every kept sensor sample
     and every trajectory sample of the compared runs, for the reference-executed T3 cases (standard / custom / white-drift
     IMUs, odometer, both frames, 100 and 200 Hz), for the given-data T1 fixtures, and for sampled runs of the REAL launches
     (65 536 and 262 144 runs, wave-specialised kernel with three producer groups, and the plain kernel).  That oracle is
     pinned to the executed reference at the stated fp32 tolerances by the CPU tests (tests/test_oracle_c.py).
  2. Against the fp64 kernel ON IDENTICAL SEEDS (the two precisions consume the same normals): per sample within the
     STATED fp32 TOLERANCES for 10 s / 1000 steps: attitude 2e-6 rad, velocity 5e-5 m/s, position 1e-4 m, accel 2e-6 m/s^2,
     gyro 1e-7 rad/s (measured: 4e-7 rad, 6e-6 m/s, 3e-5 m, 1.3e-6, 5e-8).
  3. Against the reference's fp64 goldens directly (noise-free closed loop, T3 with injected noise) at the same tolerances.
"""
import numpy as np
import pytest

from conftest import load_golden, ang_close, golden_vibration, T3_VIB

pytestmark = pytest.mark.gpu
ZERO = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, 100.0), 'arw': np.zeros(3), 'vrw': np.zeros(3)}
TOL = {'att': 2e-6, 'vel': 5e-5, 'pos_m': 1e-4, 'pos_rad': 1e-11, 'accel': 2e-6, 'gyro': 1e-7}


@pytest.fixture(scope='module')
def ctx():
    import ginsim
    c = ginsim.Context(0)
    yield c
    c.close()


def _errs(g):
    acc = {k[6:]: g[k] for k in g if k.startswith('accel_') and k != 'accel'}
    gyr = {k[5:]: g[k] for k in g if k.startswith('gyro_') and k != 'gyro'}
    return acc, gyr


def _bits_equal(dev, ora, what):
    """dev: float64 array holding widened float32 values; ora: float32 array.  Equal as float32 bit patterns (+0 == -0 aside)."""
    d32 = dev.astype(np.float32)
    assert np.array_equal(d32.astype(np.float64), dev), what + ': device values are not float32 numbers'
    bad = d32 != ora
    assert not bad.any(), '%s: %d of %d samples differ from the float oracle, worst %.3e' % (
        what, int(bad.sum()), bad.size, float(np.abs(d32[bad].astype(np.float64) - ora[bad].astype(np.float64)).max()))


def _close_to_f64(att, dpos_plus, vel, a64, p64, v64, rf, what, scale=1.0):
    assert ang_close(att, a64, TOL['att']), what + ' att'
    assert np.abs(vel - v64).max() <= TOL['vel'] * scale, what + ' vel %.2e' % np.abs(vel - v64).max()
    dp = np.abs(dpos_plus - p64)
    if rf == 1:
        assert dp.max() <= TOL['pos_m'] * scale, what + ' pos %.2e' % dp.max()
    else:
        assert dp[..., :2].max() <= TOL['pos_rad'] * scale and dp[..., 2].max() <= TOL['pos_m'] * scale, what + ' pos'


@pytest.mark.parametrize('plain', [False, True])
@pytest.mark.parametrize('name', ['t3_demo_rf1', 't3_mid_rf0', 't3_white_gps_rf0', 't3_low_rf1', 't3_high_odo_rf0', 't3_drive200_rf0'] + T3_VIB)
def test_fp32_kernel_equals_float_oracle_and_reference_t3(ctx, name, plain):
    """T3_VIB: with the vibration term of Sim(env=...) -- defined in single precision like the rest (mc_kernel_f32.hip add_vibration),
    so the float oracle reproduces it to the bit; against the reference run with env within the fp32 tolerances."""
    import ginsim
    from oracle import c_oracle
    g = load_golden(name)
    R, k, fs, rf, seed = int(g['R']), g['rows'], float(g['fs']), int(g['ref_frame']), int(g['seed'])
    acc_err, gyr_err = _errs(g)
    truth = {'ref_accel': g['ref_accel'], 'ref_gyro': g['ref_gyro'], 'ref_att': g['ref_att'], 'ref_pos': g['ref_pos'],
             'ref_vel': g['ref_vel']}
    odo_err = None
    algos = [a for a, tag in (('free', 'fi'), ('odo', 'odo')) if tag + '_att' in g]
    if 'odo' in g:
        truth['ref_odo'] = g['ref_odo']
        odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])}
    RR = 70                      # more runs than the golden holds: a ragged wavefront; the first R are the golden's
    vib_acc, vib_gyro = golden_vibration(g)
    vib = vib_acc is not None or vib_gyro is not None
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc_err, gyr_err, g['ini'], runs=RR, algos=tuple(algos), odo_err=odo_err,
                               seed=seed, keep_sensors=True, keep_traj=True, precision='f32', vib_accel=vib_acc, vib_gyro=vib_gyro)
    if plain:
        job.params.block_threads = 256
    job.run()
    assert ('split' in job.kernel_name()) == (not plain and 'free' in algos and not vib), job.kernel_name()
    assert job.kernel_name().endswith(', true>') == vib or 'split' in job.kernel_name(), job.kernel_name()
    ids = np.arange(RR)
    scale = 5.0 if g['ref_accel'].shape[0] > 1000 else 1.0
    for a, tag in (('free', 'fi'), ('odo', 'odo')):
        if a not in algos:
            continue
        end, traj, sens, odo = c_oracle.mc_run_f32(seed, 0, RR, fs, rf, truth, acc_err, gyr_err, g['ini'], algo=a, odo_err=odo_err, keep=RR,
                                                   vib_accel=vib_acc, vib_gyro=vib_gyro)
        _bits_equal(job.sensors('accel', ids), sens[:, :, 0:3], name + ' accel')
        _bits_equal(job.sensors('gyro', ids), sens[:, :, 3:6], name + ' gyro')
        if odo is not None:
            _bits_equal(job.sensors('odo', ids), odo, name + ' odo')
        att, dpos, vel = job.trajectories(a, ids, displacement=True)
        _bits_equal(att, traj[:, :, 0:3], name + a + ' att')
        _bits_equal(dpos, traj[:, :, 3:6], name + a + ' displacement')
        _bits_equal(vel, traj[:, :, 6:9], name + a + ' vel')
        dev_end = job.end_errors(a)
        assert ang_close(dev_end[:, :3], end[:, :3], 1e-12)
        np.testing.assert_allclose(dev_end[:, 3:6], end[:, 3:6], rtol=0, atol=2e-8)
        np.testing.assert_allclose(dev_end[:, 6:9], end[:, 6:9], rtol=0, atol=1e-12)
        # the reference itself (fp64, the same normals injected), first R runs, sampled rows
        att, pos, vel = job.trajectories(a, np.arange(R))
        _close_to_f64(att[:, k], pos[:, k], vel[:, k], g[tag + '_att'], g[tag + '_pos'], g[tag + '_vel'], rf, name + a, scale)
    np.testing.assert_allclose(job.sensors('accel', np.arange(R))[:, k], g['accel'], rtol=0, atol=TOL['accel'])
    np.testing.assert_allclose(job.sensors('gyro', np.arange(R))[:, k], g['gyro'], rtol=0, atol=TOL['gyro'])
    job.release()


@pytest.mark.parametrize('name', ['t3_demo_rf1', 't3_mid_rf0', 't3_high_odo_rf0'])
def test_fp32_short_series_every_tile_boundary(ctx, name):
    """The wave-specialised kernel takes the steps in tiles of six, two per loop trip, and the last sample -- sensor output
    only -- after its loops from the ring: series of 2 .. 20 samples put the end of the series on every position of a tile
    (n - 1 a multiple of six, an odd and an even number of steps in the last tile, a single step, a single tile), kept and
    not kept, against the float oracle bit for bit."""
    import ginsim
    from oracle import c_oracle
    g = load_golden(name)
    fs, rf, seed = float(g['fs']), int(g['ref_frame']), int(g['seed'])
    acc_err, gyr_err = _errs(g)
    algos = [a for a, tag in (('free', 'fi'), ('odo', 'odo')) if tag + '_att' in g]
    odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])} if 'odo' in g else None
    RR = 70
    ids = np.arange(RR)
    for n in (2, 3, 4, 6, 7, 8, 12, 13, 14, 19, 20):
        truth = {k: np.ascontiguousarray(g[k][:n]) for k in ('ref_accel', 'ref_gyro', 'ref_att', 'ref_pos', 'ref_vel')}
        if odo_err is not None:
            truth['ref_odo'] = np.ascontiguousarray(g['ref_odo'][:n])
        for keep in (True, False):
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc_err, gyr_err, g['ini'], runs=RR, algos=tuple(algos), odo_err=odo_err,
                                       seed=seed, keep_sensors=keep, keep_traj=keep, precision='f32')
            job.run()
            assert 'split' in job.kernel_name(), job.kernel_name()
            for a in algos:
                end, traj, sens, odo = c_oracle.mc_run_f32(seed, 0, RR, fs, rf, truth, acc_err, gyr_err, g['ini'], algo=a, odo_err=odo_err, keep=RR)
                tag = '%s n=%d keep=%d %s' % (name, n, keep, a)
                if keep:
                    _bits_equal(job.sensors('accel', ids), sens[:, :, 0:3], tag + ' accel')
                    _bits_equal(job.sensors('gyro', ids), sens[:, :, 3:6], tag + ' gyro')
                    if odo is not None:
                        _bits_equal(job.sensors('odo', ids), odo, tag + ' odo')
                    att, dpos, vel = job.trajectories(a, ids, displacement=True)
                    _bits_equal(att, traj[:, :, 0:3], tag + ' att')
                    _bits_equal(dpos, traj[:, :, 3:6], tag + ' displacement')
                    _bits_equal(vel, traj[:, :, 6:9], tag + ' vel')
                dev_end = job.end_errors(a)
                assert ang_close(dev_end[:, :3], end[:, :3], 1e-12), tag
                np.testing.assert_allclose(dev_end[:, 3:6], end[:, 3:6], rtol=0, atol=2e-8, err_msg=tag)
                np.testing.assert_allclose(dev_end[:, 6:9], end[:, 6:9], rtol=0, atol=1e-12, err_msg=tag)
            job.release()


@pytest.mark.parametrize('name', ['bosch', 'nxp', 'tumble'])
def test_fp32_given_data_fixtures(ctx, name):
    """The plugin boundary in fp32: logged fp64 IMU series on the device, rounded to float as the kernel reads them."""
    import ginsim
    from oracle import c_oracle
    g = load_golden('t1_fixture_' + name)
    k, fs = g['rows'], float(g['fs'])
    n = g['gyro'].shape[0]
    R = 3
    gy = ctx.upload(np.ascontiguousarray(np.repeat(g['gyro'].T[:, :, None], R, axis=2)))       # [3][n][R]
    ac = ctx.upload(np.ascontiguousarray(np.repeat(g['accel'].T[:, :, None], R, axis=2)))
    dummy = {'ref_accel': np.zeros((n, 3)), 'ref_gyro': np.zeros((n, 3)), 'ref_att': np.zeros((n, 3)), 'ref_pos': np.zeros((n, 3)),
             'ref_vel': np.zeros((n, 3))}
    for tag, rf, ini, erot in (('extg', 0, g['ini'], False), ('wgs', 0, g['ini'][:9], True), ('rf1', 1, g['ini'][:9], True)):
        job = ginsim.MonteCarloJob(ctx, fs, rf, dummy, None, None, ini, runs=R, earth_rot=erot, keep_traj=True, precision='f32',
                                   given={'gyro': gy, 'accel': ac}).run()
        assert 'mc_kernel_f32<%d, 1, true' % rf in job.kernel_name()
        att, dpos, vel = job.trajectories('free', [0, R - 1], displacement=True)
        o_att, o_dpos, o_vel, _ = c_oracle.free_integration_f32(rf, fs, g['gyro'], g['accel'], ini, earth_rot=erot)
        for r in range(2):
            _bits_equal(att[r], o_att, name + tag + ' att')
            _bits_equal(dpos[r], o_dpos, name + tag + ' displacement')
            _bits_equal(vel[r], o_vel, name + tag + ' vel')
        att, pos, vel = job.trajectories('free', [0])
        assert ang_close(att[0][k], g['att_' + tag], 1e-4 if name == 'tumble' else TOL['att'])
        _close_to_f64(g['att_' + tag], pos[0][k], vel[0][k], g['att_' + tag], g['pos_' + tag], g['vel_' + tag], rf, name + tag)
        job.release()
    gy.free()
    ac.free()


def test_fp32_given_data_both_plugins_at_other_rates(ctx):
    """Both plugins' given-data form (FreeIntegration.run(set_of_input) of free_integration.py and free_integration_odo.py) in
    fp32 at 50 / 200 / 400 Hz, both frames, external gravity / Earth rotation: the kernel equals the float oracle bit for bit."""
    import ginsim
    from oracle import c_oracle
    from test_oracle_golden import _t1_rates_cases
    bufs = {}
    for c, tag, rf, ini, erot, plug in _t1_rates_cases():
        fs, n, R = float(c['fs']), c['gyro'].shape[0], 2
        key = id(c['gyro'])
        if key not in bufs:
            bufs[key] = {'gyro': ctx.upload(np.ascontiguousarray(np.repeat(c['gyro'].T[:, :, None], R, axis=2))),
                         'accel': ctx.upload(np.ascontiguousarray(np.repeat(c['accel'].T[:, :, None], R, axis=2))),
                         'odo': ctx.upload(np.ascontiguousarray(np.repeat(c['odo'][:, None], R, axis=1)))}
        dummy = {'ref_accel': np.zeros((n, 3)), 'ref_gyro': np.zeros((n, 3)), 'ref_att': np.zeros((n, 3)), 'ref_pos': np.zeros((n, 3)),
                 'ref_vel': np.zeros((n, 3))}
        given = {'gyro': bufs[key]['gyro'], ('accel' if plug == 'free' else 'odo'): bufs[key]['accel' if plug == 'free' else 'odo']}
        job = ginsim.MonteCarloJob(ctx, fs, rf, dummy, None, None, ini, runs=R, algos=(plug,), earth_rot=erot, keep_traj=True,
                                   precision='f32', given=given).run()
        att, dpos, vel = job.trajectories(plug, [R - 1], displacement=True)
        o_att, o_dpos, o_vel, _ = c_oracle.free_integration_f32(rf, fs, c['gyro'], c['accel'] if plug == 'free' else None, ini, earth_rot=erot,
                                                                odo=c['odo'] if plug == 'odo' else None)
        what = '%s %s %g Hz' % (plug, tag, fs)
        _bits_equal(att[0], o_att, what + ' att')
        _bits_equal(dpos[0], o_dpos, what + ' displacement')
        _bits_equal(vel[0], o_vel, what + ' vel')
        job.release()
    for b in bufs.values():
        for v in b.values():
            v.free()


@pytest.mark.parametrize('R,plain', [(65536, False), (262144, False), (65536, True)])
def test_fp32_real_launch_sampled_runs(ctx, R, plain):
    """BASELINE config 5's launch at its real sizes, materialised as the bench runs it: runs drawn from the first, middle
    and last blocks equal the float oracle bit for bit, and stay within the stated tolerances of the fp64 kernel ON THE
    SAME SEEDS (same normals: the two launches differ by rounding only)."""
    import ginsim
    from ginsim import workloads
    from oracle import c_oracle
    fs, rf, seed, off = 100.0, 1, 20260923, 3 * R
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=R, seed=seed, run_offset=off, keep_sensors=True,
                               keep_traj=True, precision='f32')
    if plain:
        job.params.block_threads = 256
    job.run()
    assert ('split' in job.kernel_name()) == (not plain), job.kernel_name()
    width = 88
    blocks = [(0, width), (R // 2 - width // 2 - 5, width), (R - width, width)]
    dev_end = job.end_errors('free')
    worst = dict(att=0.0, pos=0.0, vel=0.0, accel=0.0, gyro=0.0)
    for first, count in blocks:
        ids = np.arange(first, first + count)
        end, traj, sens, _ = c_oracle.mc_run_f32(seed, off + first, count, fs, rf, truth, acc, gyr, ini, keep=count)
        att, dpos, vel = job.trajectories('free', ids, displacement=True)
        _bits_equal(job.sensors('accel', ids), sens[:, :, 0:3], 'accel')
        _bits_equal(job.sensors('gyro', ids), sens[:, :, 3:6], 'gyro')
        _bits_equal(att, traj[:, :, 0:3], 'att')
        _bits_equal(dpos, traj[:, :, 3:6], 'displacement')
        _bits_equal(vel, traj[:, :, 6:9], 'vel')
        assert ang_close(dev_end[ids, :3], end[:, :3], 1e-12)
        np.testing.assert_allclose(dev_end[ids, 3:6], end[:, 3:6], rtol=0, atol=2e-8)
        # the fp64 restatement of the same runs (the fp64 kernel equals it to 1e-9, tests/test_gpu_full_size.py)
        end64, traj64, sens64 = c_oracle.mc_run(seed, off + first, count, fs, rf, truth, acc, gyr, ini, keep=count)
        att, pos, vel = job.trajectories('free', ids)
        _close_to_f64(att, pos, vel, traj64[:, :, 0:3], traj64[:, :, 3:6], traj64[:, :, 6:9], rf, 'R=%d block %d' % (R, first))
        d_acc = np.abs(job.sensors('accel', ids) - sens64[:, :, 0:3]).max()
        d_gyr = np.abs(job.sensors('gyro', ids) - sens64[:, :, 3:6]).max()
        assert d_acc <= TOL['accel'] and d_gyr <= TOL['gyro'], (d_acc, d_gyr)
        worst = dict(att=max(worst['att'], np.abs(np.mod(att - traj64[:, :, 0:3] + np.pi, 2 * np.pi) - np.pi).max()),
                     pos=max(worst['pos'], np.abs(pos - traj64[:, :, 3:6]).max()), vel=max(worst['vel'], np.abs(vel - traj64[:, :, 6:9]).max()),
                     accel=max(worst['accel'], d_acc), gyro=max(worst['gyro'], d_gyr))
    try:
        from test_gpu_full_size import _record
        _record('c5_fp32_vs_fp64_R%d%s' % (R, '_plain' if plain else ''), **worst)
    except ImportError:
        pass
    st = job.stats('free')
    assert st.count == R
    np.testing.assert_allclose(st.std, dev_end.std(0), rtol=1e-10)
    job.release()


@pytest.mark.parametrize('rf', [0, 1])
def test_fp32_vs_fp64_kernel_identical_seeds(ctx, rf):
    """Device against device: the fp32 and the fp64 kernel on the same seeds, every run of a 4096-run launch compared at the
    end point, 64 runs per sample; the statistics of the two launches agree far inside sampling error (same noise)."""
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    R = 4096
    jobs = {}
    for prec in ('f64', 'f32'):
        jobs[prec] = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, seed=5, keep_traj=True, keep_sensors=True,
                                          precision=prec).run()
    e64, e32 = jobs['f64'].end_errors('free'), jobs['f32'].end_errors('free')
    assert ang_close(e32[:, :3], e64[:, :3], TOL['att'])
    np.testing.assert_allclose(e32[:, 6:9], e64[:, 6:9], rtol=0, atol=TOL['vel'])
    np.testing.assert_allclose(e32[:, 3:6], e64[:, 3:6], rtol=0, atol=TOL["pos_m"])
    if rf == 0:
        assert np.abs(e32[:, 3:5] - e64[:, 3:5]).max() <= TOL['pos_rad'] and np.abs(e32[:, 5] - e64[:, 5]).max() <= TOL['pos_m']
    ids = np.arange(0, R, 64)
    a32, p32, v32 = jobs['f32'].trajectories('free', ids)
    a64, p64, v64 = jobs['f64'].trajectories('free', ids)
    _close_to_f64(a32, p32, v32, a64, p64, v64, rf, 'rf%d' % rf)
    np.testing.assert_allclose(jobs['f32'].sensors('accel', ids), jobs['f64'].sensors('accel', ids), rtol=0, atol=TOL['accel'])
    np.testing.assert_allclose(jobs['f32'].sensors('gyro', ids), jobs['f64'].sensors('gyro', ids), rtol=0, atol=TOL['gyro'])
    s64, s32 = jobs['f64'].stats('free'), jobs['f32'].stats('free')
    np.testing.assert_allclose(s32.std, s64.std, rtol=2e-3)
    for j in jobs.values():
        j.release()


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
            _close_to_f64(att[r][k], pos[r][k], vel[r][k], g[tag + '_att'], g[tag + '_pos'], g[tag + '_vel'], rf, tag)
    np.testing.assert_allclose(job.sensors('gyro', [5])[0], truth['ref_gyro'], rtol=0, atol=1e-7)
    e = job.end_errors('free')
    assert np.all(e[0] == e[129])            # deterministic and lane independent
    job.release()


def test_fp32_shard_invariance(ctx):
    """Bit-identical fp32 results under any sharding of the runs over launches (as for the fp64 kernel)."""
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    whole = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=300, seed=9, precision='f32').run()
    ref = whole.end_errors('free')
    whole.release()
    first = 0
    for count in (1, 63, 65, 171):
        part = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=count, seed=9, run_offset=first, precision='f32').run()
        assert np.array_equal(part.end_errors('free'), ref[first:first + count])
        part.release()
        first += count


@pytest.mark.parametrize('name', ['t3_mid_rf0', 't3_demo_rf1', 't3_high_odo_rf0'])
def test_fp32_process_and_ned_statistics_over_float_trajectories(ctx, name):
    """process-error statistics (ins_data_manager.py:761-795) and the 'ned' position error (:542-552) of an fp32
    job.  The statistics kernel reads the float series the fp32 kernel wrote (positions as displacement from the run's initial
    position, formed back in fp64), so its result must equal the NumPy restatement applied to the host copies of those same
    series to fp64 rounding; against the reference's fp64 numbers it agrees within the fp32 trajectory tolerances."""
    import ginsim
    from oracle import ins_np
    g = load_golden(name)
    R, fs, rf = int(g['R']), float(g['fs']), int(g['ref_frame'])
    acc, gyr = _errs(g)
    truth = {'ref_accel': g['ref_accel'], 'ref_gyro': g['ref_gyro'], 'ref_pos': g['ref_pos'], 'ref_vel': g['ref_vel'], 'ref_att': g['ref_att']}
    odo_err = None
    if 'odo' in g:
        truth['ref_odo'] = g['ref_odo']
        odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])}
    algos = tuple(a for a in ('free', 'odo') if ('fi' if a == 'free' else 'odo') + '_att' in g)
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, g['ini'], runs=R, algos=algos, odo_err=odo_err, seed=int(g['seed']),
                               keep_traj=True, precision='f32').run()
    j0 = int(round(float(g['proc_start_s']) * fs))
    for a in algos:
        att, pos, vel = job.trajectories(a, np.arange(R))        # host copies: initial position + displacement
        for ned in ([False, True] if rf == 0 else [False]):
            want = ins_np.process_error_stats(att, pos, vel, g['ref_att'], g['ref_pos'], g['ref_vel'], j0, pos_ned=ned)
            got = job.process_stats(a, j0, pos_ned=ned)
            # NED metres come out of a difference of two ECEF positions (6e6 m): 1e-9 m of rounding on either side, the same
            # absolute 2e-8 m the fp64 statistics tests allow (tests/test_process_stats.py)
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=2e-8 if ned else 1e-12, err_msg='%s %s ned=%s' % (name, a, ned))
        if rf == 0:
            e = ins_np.lla_error_ned(pos[:, -1], np.broadcast_to(g['ref_pos'][-1], (R, 3)))
            end = job.stats_from_traj(a, pos_ned=True)
            np.testing.assert_allclose(end.maxabs[3:6], np.abs(e).max(0), rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(end.std[3:6], e.std(0), rtol=1e-7, atol=1e-9)
    job.release()


def test_sim_fp32_results_with_the_reference_defaults(ctx):
    """Sim(precision='f32').results() with the reference's defaults (err_stats_start = 0: per-run process statistics) -- kept
    trajectories and statistics-only (re-integrated block by block) give the same numbers, close to the fp64 Sim's."""
    import contextlib
    import io
    import os
    from conftest import PKG
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    g = load_golden('t3_mid_rf0')
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    stats = {}
    for tag, kw in (('f64', dict(precision='f64')), ('kept', dict(precision='f32', keep_trajectories=True)),
                    ('blocks', dict(precision='f32', keep_trajectories=False, max_device_bytes=9 * 4 * 1000 * 300))):
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
        sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=0, imu=imu, algorithm=free_integration.FreeIntegration(g['ini']),
                          seed=int(g['seed']), **kw)
        sim.run(700)
        with contextlib.redirect_stdout(io.StringIO()):
            sim.results(extra_opt='ned')
        st = sim.err_stats
        assert len(st['vel']['max']) == 700 and st['pos']['units'] == "['m', 'm', 'm']"
        stats[tag] = {dn: {s: np.stack([st[dn][s]['algo0_%d' % r] for r in (0, 255, 256, 699)]) for s in ('max', 'avg', 'std')}
                      for dn in ('att_euler', 'pos', 'vel')}
    for dn in ('att_euler', 'pos', 'vel'):
        for s in ('max', 'avg', 'std'):
            np.testing.assert_array_equal(stats['kept'][dn][s], stats['blocks'][dn][s])       # same runs, same kernel, same reduction
            scale = np.abs(stats['f64'][dn]['max']).max()
            np.testing.assert_allclose(stats['kept'][dn][s], stats['f64'][dn][s], rtol=0, atol=2e-3 * scale + 1e-7)


def test_sim_fp32_with_a_vibration_environment(ctx):
    """Sim(precision='f32', env=...): the vibration variants of the float kernel behind the drop-in Sim -- kept trajectories and
    statistics only (re-integrated block by block) agree with each other and, within the fp32 tolerance, with the fp64 Sim."""
    import contextlib
    import io
    import os
    from conftest import PKG
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    g = load_golden('t3_vib_sin_rf0')
    env = {k: str(g['env_' + k]) for k in ('acc', 'gyro')}
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    stats = {}
    for tag, kw in (('f64', dict(precision='f64')), ('kept', dict(precision='f32', keep_trajectories=True)),
                    ('blocks', dict(precision='f32', keep_trajectories=False, max_device_bytes=9 * 4 * 1000 * 300))):
        imu = imu_model.IMU(accuracy='low-accuracy', axis=6, gps=False)
        sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=0, imu=imu, env=env, algorithm=free_integration.FreeIntegration(g['ini']),
                          seed=int(g['seed']), **kw)
        sim.run(600)
        with contextlib.redirect_stdout(io.StringIO()):
            sim.results(err_stats_start=-1)
        stats[tag] = {dn: {s: np.asarray(sim.err_stats[dn][s]) for s in ('max', 'avg', 'std')} for dn in ('att_euler', 'pos', 'vel')}
    for dn in ('att_euler', 'pos', 'vel'):
        for s in ('max', 'avg', 'std'):
            np.testing.assert_array_equal(stats['kept'][dn][s], stats['blocks'][dn][s])
            scale = np.abs(stats['f64'][dn]['max']).max()
            np.testing.assert_allclose(stats['kept'][dn][s], stats['f64'][dn][s], rtol=0, atol=2e-3 * scale + 1e-7)
