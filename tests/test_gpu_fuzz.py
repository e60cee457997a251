"""Randomised differential test of the hot path: seeded random motion profiles (all five command types, aggressive rates so that
the pitch fold and the +-2 pi wraps of attitude.euler_update_zyx are exercised), random IMU error models (finite / infinite
correlation times, constant biases, zero axes), random vibration environments, both frames, every algorithm set, ragged run
counts, 64-bit run offsets -- device against the C restatement (oracle/c/ginsim_oracle.c) and against itself under sharding
(fp64: to SURVEY 8(c)'s 1e-9 up to the Euler singularity, see the comment in the test; fp32: bit for bit, singularity or not).
The truth comes from the native path generator (pinned against the reference by tests/test_host_cpu.py); sizes are small, the
whole file takes seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
D2R = np.pi / 180


@pytest.fixture(scope='module')
def ctx():
    import ginsim
    c = ginsim.Context(0)
    yield c
    c.close()


def _random_case(i):
    import ginsim
    rng = np.random.RandomState(7000 + i)
    rf = int(rng.randint(0, 2))
    fs = float(rng.choice([50.0, 100.0, 200.0]))
    ini = np.array([rng.uniform(-70, 70) * D2R, rng.uniform(-170, 170) * D2R, rng.uniform(0, 800), rng.uniform(0, 20), 0.0, 0.0,
                    rng.uniform(-180, 180) * D2R, rng.uniform(-20, 20) * D2R, rng.uniform(-30, 30) * D2R])
    segs = []
    for _ in range(rng.randint(1, 5)):
        t = int(rng.randint(1, 6))
        dur = float(rng.uniform(1.0, 6.0))
        if t == 1:      # rates: up to a full pitch-over within the segment
            segs.append([t, rng.uniform(-60, 60), rng.uniform(-50, 50), rng.uniform(-90, 90), rng.uniform(-2, 2), 0.0, 0.0, dur, 1])
        elif t in (2, 4):
            segs.append([t, rng.uniform(-180, 180), rng.uniform(-60, 60), rng.uniform(-120, 120), rng.uniform(0, 20) if t == 2 else rng.uniform(-4, 4),
                         0.0, 0.0, dur, 1])
        else:
            segs.append([t, rng.uniform(-170, 170), rng.uniform(-40, 40), rng.uniform(-90, 90), rng.uniform(0, 20) if t == 5 else rng.uniform(-4, 4),
                         0.0, 0.0, dur, 1])
    md = np.array(segs, dtype=np.float64)
    md[:, 1:4] *= D2R
    mob = (rng.uniform(1.0, 8.0), rng.uniform(20.0, 200.0) * D2R, rng.uniform(40.0, 300.0) * D2R)
    raw = ginsim.pathgen(ini, md, fs, 0.0, mob, rf)
    truth = {'ref_accel': np.ascontiguousarray(raw['imu'][:, 1:4]), 'ref_gyro': np.ascontiguousarray(raw['imu'][:, 4:7]),
             'ref_pos': np.ascontiguousarray(raw['nav'][:, 1:4]), 'ref_vel': np.ascontiguousarray(raw['nav'][:, 4:7]),
             'ref_att': np.ascontiguousarray(raw['nav'][:, 7:10]), 'ref_odo': np.ascontiguousarray(raw['odo'][:, 2])}

    def corr():
        c = rng.uniform(0.5, 500.0, 3)
        c[rng.rand(3) < 0.25] = np.inf
        return c

    def maybe_zero(v):
        v = np.array(v, dtype=np.float64)
        v[rng.rand(3) < 0.2] = 0.0
        return v
    acc = {'b': maybe_zero(rng.normal(0, 0.02, 3)) if rng.rand() < 0.5 else np.zeros(3), 'b_drift': maybe_zero(rng.uniform(1e-5, 5e-4, 3)),
           'b_corr': corr(), 'vrw': maybe_zero(rng.uniform(1e-4, 2e-3, 3))}
    gyr = {'b': maybe_zero(rng.normal(0, 1e-3, 3)) if rng.rand() < 0.5 else np.zeros(3), 'b_drift': maybe_zero(rng.uniform(1e-6, 1e-4, 3)),
           'b_corr': corr(), 'arw': maybe_zero(rng.uniform(1e-5, 5e-4, 3))}

    def vib(scale):
        k = rng.randint(0, 3)
        if k == 0:
            return None
        v = {'type': 'random' if k == 1 else 'sinusoidal', 'x': rng.uniform(0, scale), 'y': rng.uniform(0, scale), 'z': rng.uniform(0, scale)}
        if k == 2:
            v['freq'] = rng.uniform(0.1, 0.45 * fs)
        return v
    algos = [('free',), ('odo',), ('free', 'odo')][rng.randint(0, 3)]
    return dict(rf=rf, fs=fs, ini=ini, truth=truth, acc=acc, gyr=gyr, va=vib(0.5), vg=vib(5e-3), algos=algos,
                odo_err={'scale': rng.uniform(0.99, 1.01), 'stdv': rng.uniform(0.0, 0.2)}, runs=int(rng.randint(1, 200)),
                off=int(rng.choice([0, 12345, 2 ** 40 + 17])), seed=int(rng.randint(0, 2 ** 62)), earth_rot=bool(rng.randint(0, 2)))


_WORST = {}


@pytest.mark.parametrize('i', range(40))
def test_random_configuration_against_the_c_oracle(ctx, i):
    import ginsim
    from oracle import c_oracle
    c = _random_case(i)
    kw = dict(algos=c['algos'], odo_err=c['odo_err'], earth_rot=c['earth_rot'], seed=c['seed'], vib_accel=c['va'], vib_gyro=c['vg'])
    R, off = c['runs'], c['off']
    job = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], runs=R, run_offset=off, keep_sensors=True,
                               keep_traj=True, **kw).run()
    keep = min(R, 3)
    n = c['truth']['ref_accel'].shape[0]
    for a in c['algos']:
        end, traj, sens = c_oracle.mc_run(c['seed'], off, R, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], algo=a, odo_err=c['odo_err'],
                                          earth_rot=c['earth_rot'], keep=keep, vib_accel=c['va'], vib_gyro=c['vg'])
        ids = np.arange(keep)
        np.testing.assert_allclose(job.sensors('accel', ids), sens[:, :, 0:3], rtol=0, atol=2e-12, err_msg='case %d accel' % i)
        np.testing.assert_allclose(job.sensors('gyro', ids), sens[:, :, 3:6], rtol=0, atol=2e-14, err_msg='case %d gyro' % i)
        att, pos, vel = job.trajectories(a, ids)
        # The Euler-rate integration of the reference (attitude.py:679-721) divides by cos(pitch): a run that passes the
        # singularity amplifies rounding differences without bound (the NumPy and the C restatement then differ from EACH OTHER:
        # 5e-4 rad in case 23, min |cos pitch| 8e-5) and may take the pitch fold on one side only.  Such a run is compared up to
        # the sample where |cos pitch| first drops below 0.02; what follows is as (in)accurate in every implementation.
        # Tolerance: SURVEY 8(c)'s 1e-9 max(1, |x|) per 1000 steps on every compared sample (measured worst over the 168 runs of this
        # file: 6.4e-14 for runs that keep |cos pitch| >= 0.2, 7.5e-14 for those that come closer -- gpurun_out/parity_margins.json).
        for r in range(keep):
            near = np.where(np.abs(np.cos(traj[r, :, 1])) < 0.02)[0]
            upto = int(near[0]) if near.size else n
            if upto < 2:
                continue
            benign = np.abs(np.cos(traj[r, :upto, 1])).min() >= 0.2
            tol = 1e-9 * max(1.0, n / 1000.0)
            d = np.mod(att[r, :upto] - traj[r, :upto, 0:3] + np.pi, 2 * np.pi) - np.pi
            assert np.abs(d).max() < tol, 'case %d %s run %d attitude %.3e (compared %d of %d samples)' % (i, a, r, np.abs(d).max(), upto, n)
            scale = np.maximum(1.0, np.abs(traj[r, :upto, 3:9]))
            err = np.abs(np.concatenate([pos[r, :upto], vel[r, :upto]], axis=1) - traj[r, :upto, 3:9]) / scale
            assert err.max() < tol, 'case %d %s run %d pos/vel %.3e (compared %d of %d samples)' % (i, a, r, err.max(), upto, n)
            key = 'benign' if benign else 'near_singular'
            _WORST[key] = max(_WORST.get(key, 0.0), float(max(np.abs(d).max(), err.max()) / max(1.0, n / 1000.0)))
            _WORST['runs_' + key] = _WORST.get('runs_' + key, 0) + 1
        from test_gpu_full_size import _record
        _record('fuzz_fp64_vs_c_oracle_per_1000_steps', **_WORST)
    # sharding: the same global runs as two launches, nothing kept -> the same end-point errors to the bit
    if R >= 2:
        cut = R // 3 + 1
        parts = [ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], runs=r, run_offset=off + o, **kw).run()
                 for o, r in ((0, cut), (cut, R - cut))]
        for a in c['algos']:
            assert np.array_equal(np.concatenate([q.end_errors(a) for q in parts]), job.end_errors(a)), 'case %d shard' % i
        for q in parts:
            q.release()
    job.release()


@pytest.mark.parametrize('i', range(80, 96))
def test_random_psd_vibration_against_the_numpy_oracle(ctx, i):
    """Sim(env=<PSD array>) (ABI 8): random PSDs -- a few rows to interpolate, rows beyond fs / 2 (the reference then returns zeros:
    time_series_from_psd.py:32-34), or a PSD GIVEN on the series' own grid (halved in place by the reference at every run, so run g
    sees 0.5^(g + 1): nothing is left of it at a run offset of 2^40) -- on the random profiles, IMU models and run offsets of this
    file, odd and even series lengths: sensors per sample against the NumPy restatement of time_series_from_psd (np.fft.ifft, as the
    reference; the goldens pin it), trajectories against its mechanisation."""
    import ginsim
    from oracle import ins_np
    c = _random_case(i)
    rng = np.random.RandomState(9000 + i)
    n, fs = c['truth']['ref_accel'].shape[0], c['fs']

    def psd(scale):
        kind = rng.randint(0, 4)
        if kind == 0:
            return None
        if kind == 3:                   # on the grid of this series
            L = min(n + n % 2, 16384) // 2 + 1
            f = np.linspace(0.0, fs / 2.0, L)
        else:
            f = np.sort(rng.uniform(0.0, (0.5 if kind == 1 else 0.7) * fs, rng.randint(2, 9)))
        return {'type': 'psd', 'freq': f, 'x': scale * rng.uniform(0, 1, f.shape), 'y': scale * rng.uniform(0, 1, f.shape), 'z': scale * rng.uniform(0, 1, f.shape)}
    va, vg = psd(1e-2), psd(1e-6)
    R, off = min(c['runs'], 70), c['off']
    kw = dict(algos=c['algos'], odo_err=c['odo_err'], earth_rot=c['earth_rot'], seed=c['seed'], vib_accel=va, vib_gyro=vg)
    job = ginsim.MonteCarloJob(ctx, fs, c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], runs=R, run_offset=off, keep_sensors=True,
                               keep_traj=True, **kw).run()
    ids = np.unique([0, R // 2, R - 1])
    a_ref, g_ref = ins_np.mc_sensors(c['seed'], off + ids, fs, c['truth']['ref_accel'], c['truth']['ref_gyro'], c['acc'], c['gyr'], va, vg)
    np.testing.assert_allclose(job.sensors('accel', ids), a_ref, rtol=0, atol=2e-12, err_msg='case %d accel' % i)
    np.testing.assert_allclose(job.sensors('gyro', ids), g_ref, rtol=0, atol=2e-14, err_msg='case %d gyro' % i)
    for v, sensor, ref in ((va, 'accel', a_ref), (vg, 'gyro', g_ref)):
        if v is not None and ginsim.psd_amplitudes(v, fs, n) is None:          # beyond fs / 2: the launch without an environment
            plain = ins_np.mc_sensors(c['seed'], off + ids, fs, c['truth']['ref_accel'], c['truth']['ref_gyro'], c['acc'], c['gyr'])[sensor == 'gyro']
            assert np.array_equal(ref, plain)
    odo = ins_np.mc_odo(c['seed'], off + ids, c['truth']['ref_odo'], c['odo_err']) if 'odo' in c['algos'] else None
    for a in c['algos']:
        att, pos, vel = ins_np.free_integration(c['rf'], fs, g_ref, a_ref, c['ini'], odo=odo if a == 'odo' else None, earth_rot=c['earth_rot'])
        got = job.trajectories(a, ids)
        for k in range(len(ids)):
            near = np.where(np.abs(np.cos(att[k, :, 1])) < 0.02)[0]
            upto = int(near[0]) if near.size else n
            if upto < 2:
                continue
            tol = 1e-9 * max(1.0, n / 1000.0)
            d = np.mod(got[0][k, :upto] - att[k, :upto] + np.pi, 2 * np.pi) - np.pi
            assert np.abs(d).max() < tol, 'case %d %s run %d attitude %.3e' % (i, a, ids[k], np.abs(d).max())
            ref = np.concatenate([pos[k, :upto], vel[k, :upto]], axis=1)
            err = np.abs(np.concatenate([got[1][k, :upto], got[2][k, :upto]], axis=1) - ref) / np.maximum(1.0, np.abs(ref))
            assert err.max() < tol, 'case %d %s run %d pos/vel %.3e' % (i, a, ids[k], err.max())
    job.release()


@pytest.mark.parametrize('i', range(40, 52))
def test_random_configuration_fp32_against_the_float_oracle(ctx, i):
    """The same generator through the single-precision kernels: bit for bit against the float restatement."""
    import ginsim
    from oracle import c_oracle
    c = _random_case(i)
    algos = tuple(a for a in c['algos'])
    R, off = min(c['runs'], 70), c['off']
    job = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], runs=R, run_offset=off, keep_sensors=True,
                               keep_traj=True, precision='f32', algos=algos, odo_err=c['odo_err'], earth_rot=c['earth_rot'], seed=c['seed'],
                               vib_accel=c['va'], vib_gyro=c['vg']).run()
    ids = np.arange(R)
    for a in algos:
        end, traj, sens, odo = c_oracle.mc_run_f32(c['seed'], off, R, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], algo=a,
                                                   odo_err=c['odo_err'], earth_rot=c['earth_rot'], keep=R, vib_accel=c['va'], vib_gyro=c['vg'])
        for name, dev, ora in (('accel', job.sensors('accel', ids), sens[:, :, 0:3]), ('gyro', job.sensors('gyro', ids), sens[:, :, 3:6])):
            assert np.array_equal(dev.astype(np.float32), ora), 'case %d %s: %d samples differ' % (i, name, int((dev.astype(np.float32) != ora).sum()))
        att, dpos, vel = job.trajectories(a, ids, displacement=True)
        for name, dev, ora in (('att', att, traj[:, :, 0:3]), ('displacement', dpos, traj[:, :, 3:6]), ('vel', vel, traj[:, :, 6:9])):
            bad = dev.astype(np.float32) != ora
            assert not bad.any(), 'case %d %s %s: %d of %d samples differ' % (i, a, name, int(bad.sum()), bad.size)
    job.release()


@pytest.mark.parametrize('i', range(60, 76))
def test_random_configuration_statistics_replay_and_series(ctx, i):
    """The other launch forms on the same random configurations: online process statistics (statistics-only kernels) against the
    kernel that reads kept trajectories and against the NumPy restatement of InsDataMgr.__process_error_stats; the given-sensors
    replay (the plugin boundary for a whole batch) bit for bit; the time-parallel series path against the lane-per-run kernels."""
    import ginsim
    from oracle import ins_np
    c = _random_case(i)
    rng = np.random.RandomState(9000 + i)
    a = c['algos'][int(rng.randint(0, len(c['algos'])))]
    R, off = c['runs'], c['off']
    n = c['truth']['ref_accel'].shape[0]
    kw = dict(odo_err=c['odo_err'], earth_rot=c['earth_rot'], seed=c['seed'], vib_accel=c['va'], vib_gyro=c['vg'])
    kept = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], runs=R, run_offset=off, algos=(a,),
                                keep_sensors=True, keep_traj=True, **kw).run()
    first = int(rng.randint(0, n - 1))
    ned = bool(c['rf'] == 0 and rng.randint(0, 2))
    online = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], c['truth'], c['acc'], c['gyr'], c['ini'], runs=R, run_offset=off, algos=(a,),
                                  proc_first=first, proc_ned=ned, end_ned=(c['rf'] == 0), **kw).run()
    assert np.array_equal(online.end_errors(a), kept.end_errors(a)), 'case %d: statistics-only launch, other trajectories' % i
    st = online.process_stats_online(a)
    st_kept = kept.process_stats(a, first, pos_ned=ned)
    # raw sums (online) against Welford (kept): the floor of the raw-sum form is 1.5e-8 |mean| on the std
    # NED metres are a difference of ECEF coordinates (6e6 m: an ulp is 1e-9 m) formed by two different instruction sequences
    ecef = np.zeros(9)
    if ned:
        ecef[3:6] = 2e-8
    floor = 3e-8 * np.abs(st_kept[:, 1]) + 1e-13 + ecef
    assert np.all(np.abs(st[:, 0] - st_kept[:, 0]) <= 1e-12 * st_kept[:, 0] + 1e-15 + ecef), 'case %d max' % i
    assert np.all(np.abs(st[:, 1] - st_kept[:, 1]) <= 1e-9 * np.abs(st_kept[:, 1]) + 1e-13 + ecef), 'case %d mean' % i
    assert np.all(np.abs(st[:, 2] - st_kept[:, 2]) <= 1e-7 * st_kept[:, 2] + floor), 'case %d std' % i
    ids = np.arange(min(R, 4))
    att, pos, vel = kept.trajectories(a, ids)
    want = ins_np.process_error_stats(att, pos, vel, c['truth']['ref_att'], c['truth']['ref_pos'], c['truth']['ref_vel'], first, pos_ned=ned)
    np.testing.assert_allclose(st_kept[ids], want, rtol=1e-8, atol=1e-12 + (2e-8 if ned else 0.0), err_msg='case %d kept statistics vs NumPy' % i)
    # the plugin boundary for a batch: FreeIntegration.run on the device-resident sensors of `kept`
    given = {'gyro': kept.buffer('gyro')}
    if a == 'free':
        given['accel'] = kept.buffer('accel')
    else:
        given['odo'] = kept.buffer('odo')
    rep = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], c['truth'], None, None, c['ini'], runs=R, algos=(a,), earth_rot=c['earth_rot'],
                               keep_traj=True, given=given).run()
    for x, y in zip(rep.trajectories(a, ids), (att, pos, vel)):
        assert np.array_equal(x, y), 'case %d: replay of the materialised sensors differs' % i
    assert np.array_equal(rep.end_errors(a), kept.end_errors(a))
    # sensors only, few runs, the truth tiled to a long series: time-parallel kernels against one lane per run
    reps = int(np.ceil(2300.0 / n))
    long_truth = {k: (np.tile(v, (reps, 1)) if v.ndim == 2 else np.tile(v, reps)) for k, v in c['truth'].items()}
    few = int(rng.randint(1, 6))
    skw = dict(algos=(), odo_err=c['odo_err'], seed=c['seed'], vib_accel=c['va'], vib_gyro=c['vg'], keep_sensors=True)
    ser = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], long_truth, c['acc'], c['gyr'], None, runs=few, run_offset=off, **skw).run()
    lane = ginsim.MonteCarloJob(ctx, c['fs'], c['rf'], long_truth, c['acc'], c['gyr'], None, runs=1030, run_offset=off, **skw).run()
    assert ser.sensor_layout == 'series' and lane.sensor_layout == 'runs'
    sid = np.arange(few)
    sa = np.abs(long_truth['ref_accel']).max() + 1.0
    np.testing.assert_allclose(ser.sensors('accel', sid), lane.sensors('accel', sid), rtol=0, atol=4e-16 * sa * 8, err_msg='case %d series accel' % i)
    np.testing.assert_allclose(ser.sensors('gyro', sid), lane.sensors('gyro', sid), rtol=0, atol=2e-15, err_msg='case %d series gyro' % i)
    np.testing.assert_allclose(ser.sensors('odo', sid), lane.sensors('odo', sid), rtol=0, atol=1e-13, err_msg='case %d series odo' % i)
    for q in (kept, online, rep, ser, lane):
        q.release()
