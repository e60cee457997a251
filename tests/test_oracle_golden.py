"""Pin the NumPy oracle (oracle/ins_np.py, oracle/philox.py) against golden vectors produced by
EXECUTING the unmodified reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import ins_np, philox
from conftest import load_golden, assert_traj_close, ang_close, golden_vibration, T3_VIB, T3_PSD


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 with 7 rounds (the engine's generator) and with 10 (Random123's default)
    kat = {7: [((0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
               ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
               ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
                (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a))],
           10: [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
                ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
                ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
                 (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]}
    assert philox.ROUNDS == 7
    for rounds, vectors in kat.items():
        for c, k, want in vectors:
            got = philox.philox4x32(*[np.uint64(x) for x in c], k[0], k[1], rounds=rounds)
            assert tuple(int(x) for x in got) == want


def test_normal_table_is_committed_and_accurate():
    """The coefficient table is part of the definition of the stream: the oracle reads the committed file the device and the
    C oracle compile in.  Its cubics must invert the normal tail probability: |z - Phi^-1(1 - t)| <= 6e-7 on every segment
    (tools/gen_normal_tables.py), i.e. one float ulp at |z| ~ 6."""
    import os
    import re
    from scipy.special import ndtri
    from conftest import PKG
    txt = open(os.path.join(PKG, 'csrc', 'normal_tables.inc')).read()
    w = np.array([int(x, 16) for x in re.findall(r'0x([0-9a-f]{8})u', txt)], dtype=np.uint32)
    assert w.size == 31 * 8 * 4
    assert np.array_equal(w.view(np.float32).reshape(248, 4), philox.normal_tables())
    # every segment: both edges, the words next to them and a sweep inside (magnitudes m = ((8 + sub) << 27 | frac) >> lz)
    rng = np.random.RandomState(3)
    worst = 0.0
    for lz in range(1, 32):
        for sub in range(8):
            frac = np.concatenate([[0, 1, 2 ** 28 - 1, 2 ** 28 - 2], rng.randint(0, 2 ** 28, 400)]).astype(np.uint64)
            y = (np.uint64(1) << np.uint64(31)) | (np.uint64(sub) << np.uint64(28)) | frac
            m = (y >> np.uint64(lz)) | np.uint64(1)
            z = philox.normal_icdf(m).astype(np.float64)
            worst = max(worst, np.abs(z + ndtri(m.astype(np.float64) * 2.0 ** -32)).max())
    assert worst < 6e-7, worst
    # known answers of the transform itself (word -> normal), exact single-precision numbers; sign = bit 31
    z = philox.normal_icdf(np.array([0, 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 0x40000000, 0x3fffffff, 12345], dtype=np.uint64))
    assert z.dtype == np.float32
    np.testing.assert_allclose(z[[0, 1, 3]], [6.2302604, 6.2302604, -6.2302604], rtol=1e-7)      # m = 1: t = 2^-32
    np.testing.assert_allclose(z[[5, 6]], [0.67448956, 0.6744898], rtol=2e-7)                      # t = 1/4: the quartile
    assert 0 < z[2] < 1e-7 and -1e-7 < z[4] < 0


def test_normals_moments():
    z0, z1 = philox.normal_pair(99, np.arange(8)[None, :], 3, np.arange(50000)[:, None])
    for z in (z0, z1):
        assert abs(z.mean()) < 4 / np.sqrt(z.size)
        assert abs(z.std() - 1) < 4 / np.sqrt(2 * z.size)
    assert abs(np.mean(z0 * z1)) < 4 / np.sqrt(z0.size)


def test_stream_cut_statistics():
    """Two normal pairs are cut from one Philox block (oracle/philox.py): the six normals of three consecutive streams
    (two of them halves of the same block) must be standard normal (Kolmogorov-Smirnov), mutually uncorrelated -- also in
    their squares, which would expose shared radius bits -- and uncorrelated along the sample index and across runs."""
    from scipy import stats
    n = 200000
    j = np.arange(n, dtype=np.uint64)
    z = []
    for stream in (3, 4, 5):                      # block 1 half 1, block 2 halves 0 and 1
        a, b = philox.normal_pair(12345, 77, stream, j)
        z += [a, b]
    z = np.array(z)
    for k in range(6):
        assert stats.kstest(z[k], 'norm').pvalue > 1e-3, 'normal %d of the group fails KS' % k
        assert abs(z[k].mean()) < 4.5 / np.sqrt(n) and abs(z[k].var() - 1.0) < 4.5 * np.sqrt(2.0 / n)
        assert abs(stats.kurtosis(z[k])) < 4.5 * np.sqrt(24.0 / n)
    lim = 4.5 / np.sqrt(n)
    c = np.corrcoef(z)
    c2 = np.corrcoef(z ** 2)
    for a in range(6):
        for b in range(a + 1, 6):
            assert abs(c[a, b]) < lim and abs(c2[a, b]) < lim, (a, b, c[a, b], c2[a, b])
        assert abs(np.corrcoef(z[a][:-1], z[a][1:])[0, 1]) < lim            # consecutive samples
    other = philox.normal_pair(12345, 78, 5, j)[0]                             # the neighbouring run
    assert abs(np.corrcoef(z[4], other)[0, 1]) < lim
    # the two words of a half block are independent uniforms
    w = philox.stream_words(12345, 77, 4, j)
    u1 = (w[0].astype(np.float64) + 0.5) * 2.0 ** -32
    u2 = (w[1].astype(np.float64) + 0.5) * 2.0 ** -32
    assert abs(np.corrcoef(u1, u2)[0, 1]) < lim
    assert stats.kstest(u1, 'uniform').pvalue > 1e-3 and stats.kstest(u2, 'uniform').pvalue > 1e-3
    # the single-precision inversion against scipy's inverse CDF in double precision on the same words
    from scipy.special import ndtri
    m = (w[0] & np.uint64(0x7fffffff)) | np.uint64(1)
    want = -ndtri(m.astype(np.float64) * 2.0 ** -32) * np.where(w[0] >> np.uint64(31), -1.0, 1.0)
    assert np.abs(philox.normal_icdf(w[0]).astype(np.float64) - want).max() < 6e-7


def test_normal_distribution_of_the_generator():
    """Distribution of the generator itself: 2e7 draws of four streams -- mean, variance, skewness, kurtosis
    within 4.5 sigma of their sampling error, tail counts beyond 3 / 4 / 5 sigma against erfc (Poisson limits), no draw beyond
    the 6.23-sigma bound of a 31-bit magnitude, and no correlation between streams, between the words of a block, or along
    the sample index (lags 1, 2, 3 and 12 = one IMU step apart in the consumption order)."""
    from scipy import stats
    from scipy.special import erfc
    n = 5000000
    j = np.arange(n, dtype=np.uint64)
    zs = []
    for stream in (0, 1, 4, 5):
        a, b = philox.normal_pair(20260924, 5, stream, j)
        zs += [a, b]
    z = np.concatenate(zs)
    N = z.size
    assert abs(z.mean()) < 4.5 / np.sqrt(N)
    assert abs(z.var() - 1.0) < 4.5 * np.sqrt(2.0 / N)
    assert abs(stats.skew(z)) < 4.5 * np.sqrt(6.0 / N)
    assert abs(stats.kurtosis(z)) < 4.5 * np.sqrt(24.0 / N)
    assert np.abs(z).max() <= 6.2302604
    for k in (3.0, 4.0, 5.0):
        expect = N * erfc(k / np.sqrt(2.0))
        got = int((np.abs(z) > k).sum())
        assert abs(got - expect) < 4.5 * np.sqrt(expect) + 1, (k, got, expect)
    lim = 4.5 / np.sqrt(n)
    zz = np.array(zs)
    c = np.corrcoef(zz)
    assert np.abs(c - np.eye(8)).max() < lim
    for lag in (1, 2, 3, 12):
        assert abs(np.mean(zz[0][:-lag] * zz[0][lag:])) < lim
    assert stats.kstest(z[::40], 'norm').pvalue > 1e-3


@pytest.mark.parametrize('name', ['bosch', 'nxp', 'tumble'])
def test_t1_given_data_fixture(name):
    g = load_golden('t1_fixture_' + name)
    k = g['rows']
    gyro, accel = g['gyro'][None], g['accel'][None]
    for tag, rf, ini, erot in (('extg', 0, g['ini'], False), ('wgs', 0, g['ini'][:9], True),
                               ('rf1', 1, g['ini'][:9], True)):
        att, pos, vel = ins_np.free_integration(rf, float(g['fs']), gyro, accel, ini, earth_rot=erot)
        assert_traj_close(att[0][k], pos[0][k], vel[0][k], g['att_' + tag], g['pos_' + tag],
                          g['vel_' + tag], rtol=1e-11, what=name + tag)


@pytest.mark.parametrize('rf', [0, 1])
def test_t2_pathgen_and_noise_free_loop(rf):
    g = load_golden('t2_turn_rf%d' % rf)
    k = g['rows']
    r = ins_np.path_gen(g['ini_pva'], g['motion_def'], float(g['fs']), float(g['fs_gps']),
                        g['mobility'], rf, gps=True, odo=True)
    assert r['imu'].shape[0] == int(g['n'])
    np.testing.assert_allclose(r['imu'][:, 1:4], g['full_ref_accel'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r['imu'][:, 4:7], g['full_ref_gyro'], rtol=0, atol=1e-14)
    np.testing.assert_allclose(r['nav'][k, 1:4], g['ref_pos'], rtol=1e-14, atol=0)
    np.testing.assert_allclose(r['nav'][k, 4:7], g['ref_vel'], rtol=0, atol=1e-12)
    assert ang_close(r['nav'][k, 7:10], g['ref_att'], 1e-13)
    np.testing.assert_allclose(r['gps'][:, 1:7], g['ref_gps'], rtol=1e-14, atol=1e-12)
    np.testing.assert_allclose(r['odo'][k, 2], g['ref_odo'], rtol=0, atol=1e-12)
    ini = g['ini_pva']
    gy, ac = r['imu'][None, :, 4:7], r['imu'][None, :, 1:4]
    att, pos, vel = ins_np.free_integration(rf, 100.0, gy, ac, ini)
    assert_traj_close(att[0][k], pos[0][k], vel[0][k], g['fi_att'], g['fi_pos'], g['fi_vel'],
                      rtol=1e-11, what='fi')
    att, pos, vel = ins_np.free_integration(rf, 100.0, gy, ac, ini, odo=r['odo'][None, :, 2])
    assert_traj_close(att[0][k], pos[0][k], vel[0][k], g['odo_att'], g['odo_pos'], g['odo_vel'],
                      rtol=1e-11, what='odo')


@pytest.mark.parametrize('rf', [0, 1])
def test_oracle_pathgen_every_command_type(rf):
    g = load_golden('truth_mixed_types_rf%d' % rf)
    k, kg = g['rows'], g['gps_rows']
    r = ins_np.path_gen(g['ini_pva'], g['motion_def'], float(g['fs']), float(g['fs_gps']), g['mobility'], rf, gps=True, odo=True)
    assert r['imu'].shape[0] == int(g['n']) and r['gps'].shape[0] == int(g['m'])
    np.testing.assert_allclose(r['imu'][k], g['imu'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r['nav'][k, 1:4], g['nav'][:, 1:4], rtol=1e-14, atol=0)
    np.testing.assert_allclose(r['nav'][k, 4:7], g['nav'][:, 4:7], rtol=0, atol=1e-12)
    assert ang_close(r['nav'][k, 7:10], g['nav'][:, 7:10], 1e-13)
    np.testing.assert_allclose(r['gps'][kg, 1:7], g['gps'][:, 1:7], rtol=1e-14, atol=1e-12)


def _errs(g):
    acc = {k[6:]: g[k] for k in g if k.startswith('accel_') and k != 'accel'}
    gyr = {k[5:]: g[k] for k in g if k.startswith('gyro_') and k != 'gyro'}
    return acc, gyr


@pytest.mark.parametrize('name', ['t3_demo_rf1', 't3_mid_rf0', 't3_white_gps_rf0', 't3_low_rf1', 't3_high_odo_rf0', 't3_drive200_rf0'] + T3_VIB + T3_PSD)
def test_t3_injected_noise_end_to_end(name):
    """T3_VIB: the same with Sim(env=...) -- random and sinusoidal vibration models (pathgen.py:476-492, 538-556).  T3_PSD: env as an
    (n, 4) PSD array (:479-484, :541-546 -> time_series_from_psd.py): interpolated to the series' grid, given on it (the reference then
    halves the caller's array in place at every run), an odd series length, a series tiled beyond 16384 samples."""
    g = load_golden(name)
    R, k, fs, rf = int(g['R']), g['rows'], float(g['fs']), int(g['ref_frame'])
    acc_err, gyr_err = _errs(g)
    runs = np.arange(R)
    vib_acc, vib_gyro = golden_vibration(g)
    assert (name in T3_VIB + T3_PSD) == (vib_acc is not None or vib_gyro is not None)
    accel, gyro = ins_np.mc_sensors(int(g['seed']), runs, fs, g['ref_accel'], g['ref_gyro'], acc_err, gyr_err, vib_acc, vib_gyro)
    np.testing.assert_allclose(accel[:, k], g['accel'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(gyro[:, k], g['gyro'], rtol=0, atol=1e-14)
    odo = None
    if 'odo' in g:
        odo = ins_np.mc_odo(int(g['seed']), runs, g['ref_odo'],
                            {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])})
        np.testing.assert_allclose(odo[:, k], g['odo'], rtol=0, atol=1e-12)
    if 'gps' in g:
        z = [philox.gps_normals(int(g['seed']), r, g['ref_gps'].shape[0]) for r in runs]
        gps = ins_np.gps_errors(g['ref_gps'], {'stdp': g['gps_stdp'], 'stdv': g['gps_stdv']}, rf,
                                np.stack([a for a, _ in z]), np.stack([b for _, b in z]))
        np.testing.assert_allclose(gps, g['gps'], rtol=1e-14, atol=1e-12)
    algos = [a for a in ('odo', 'fi') if a + '_att' in g]
    order = sorted(algos, key=lambda a: 0)  # stats keys are algo0/algo1 in list order of the case
    for a in algos:
        att, pos, vel = ins_np.free_integration(rf, fs, gyro, accel, g['ini'], odo=odo if a == 'odo' else None)
        assert_traj_close(att[:, k], pos[:, k], vel[:, k], g[a + '_att'], g[a + '_pos'], g[a + '_vel'],
                          rtol=1e-10, what=name + a)
        e = ins_np.end_point_errors(att, pos, vel, g['ref_att'], g['ref_pos'], g['ref_vel'])
        st = ins_np.array_stats(e)
        # which group is this algo? match by comparing to every stored group
        groups = sorted({key.rsplit('_', 1)[1] for key in g if key.startswith('stat_att_euler_max_')})
        r2d = 180.0 / np.pi
        scale = {'att_euler': np.full(3, r2d), 'pos': np.array([r2d, r2d, 1.0]) if rf == 0 else np.ones(3),
                 'vel': np.ones(3)}
        ok = False
        for grp in groups:
            good = True
            for dn, sl in (('att_euler', slice(0, 3)), ('pos', slice(3, 6)), ('vel', slice(6, 9))):
                for s in ('max', 'avg', 'std'):
                    want = g['stat_%s_%s_%s' % (dn, s, grp)]
                    got = st[s][sl] * scale[dn]
                    good &= bool(np.allclose(got, want, rtol=1e-7, atol=1e-12))
            ok |= good
        assert ok, 'end-point statistics of %s/%s match no reference group' % (name, a)


@pytest.mark.parametrize('name,grade', [('t3_mid_rf0', 'mid-accuracy'), ('t3_low_rf1', 'low-accuracy'),
                                        ('t3_high_odo_rf0', 'high-accuracy'), ('t3_drive200_rf0', 'low-accuracy')])
def test_t3_goldens_hold_the_grade_their_name_claims(name, grade):
    """The reference's IMU(accuracy=dict) overwrites the module-level 'low-accuracy' dicts
    (gnss_ins_sim/sim/imu_model.py:110-112, 138-158): generated in ONE interpreter after a dict case, a 'low-accuracy'
    golden silently holds the dict's IMU (round 2's t3_low_rf1 did).  make_golden.py now runs every case in a fresh
    interpreter; this test pins the recorded parameters to the grade table (imu_model.py:18-52 values)."""
    from gnss_ins_sim.sim import imu_model
    g = load_golden(name)
    imu = imu_model.IMU(accuracy=grade, axis=6, gps=False)
    for k, v in imu.gyro_err.items():
        np.testing.assert_array_equal(g['gyro_' + k], v, err_msg='%s gyro %s' % (name, k))
    for k, v in imu.accel_err.items():
        np.testing.assert_array_equal(g['accel_' + k], v, err_msg='%s accel %s' % (name, k))
    # and the three grades really differ
    other = imu_model.IMU(accuracy='mid-accuracy' if grade != 'mid-accuracy' else 'low-accuracy', axis=6, gps=False)
    assert not np.array_equal(g['gyro_arw'], other.gyro_err['arw'])


def test_golden_recipe_lists_every_committed_file():
    """Every .npz under tests/golden/ is produced by a case of make_golden.py (so `--check` covers all of them)."""
    import ast
    import os
    from conftest import GOLDEN
    src = open(os.path.join(GOLDEN, 'make_golden.py')).read()
    tree = ast.parse(src)
    cases = next(n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], 'id', '') == 'CASES')
    listed = {f for c in ast.literal_eval(cases.value) for f in c[2]}
    have = {f for f in os.listdir(GOLDEN) if f.endswith('.npz')}
    assert have == {f for f in listed if f.endswith('.npz')}


def test_t2_long_drive_truth_rows():
    """Full-length pathgen restatement is slow in Python (193k steps); run it only on request."""
    import os
    if not os.environ.get('GINSIM_SLOW'):
        pytest.skip('set GINSIM_SLOW=1 (the C oracle covers this case in test_oracle_c.py)')
    g = load_golden('t2_long_drive_rf0')
    r = ins_np.path_gen(g['ini_pva'], g['motion_def'], float(g['fs']), float(g['fs_gps']), g['mobility'], 0,
                        gps=True, odo=True)
    assert r['imu'].shape[0] == int(g['n']) and r['gps'].shape[0] == int(g['m'])
    np.testing.assert_allclose(r['imu'][g['rows']], g['imu'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r['nav'][g['rows'], 1:7], g['nav'][:, 1:7], rtol=1e-12, atol=1e-9)


def test_allan_matches_reference():
    g = load_golden('allan_ref')
    n, fs, seed = int(g['n']), float(g['fs']), int(g['seed'])
    j = np.arange(n, dtype=np.uint64)
    x = 0.3 * philox.normal_pair(seed, 7, 5, j)[0] + 1e-3 * np.cumsum(philox.normal_pair(seed, 7, 4, j)[1])
    avar, tau = ins_np.allan_var(x, fs)
    np.testing.assert_allclose(tau, g['tau'], rtol=0, atol=0)
    np.testing.assert_allclose(avar, g['avar'], rtol=1e-12)


@pytest.mark.parametrize('rf', [0, 1])
def test_t3_magnetometer_and_gps_models(rf):
    """9-axis + GPS: truth magnetometer (pathgen with the stored WMM field), mag_gen with soft/hard iron, gps_gen."""
    g = load_golden('t3_mag9_gps_rf%d' % rf)
    t2 = load_golden('t2_turn_rf%d' % rf)
    r = ins_np.path_gen(t2['ini_pva'], t2['motion_def'], 100.0, 10.0, t2['mobility'], rf, gps=True, geo_mag_n=g['geo_mag_n'])
    np.testing.assert_allclose(r['mag'][:, 1:4], g['ref_mag'], rtol=0, atol=1e-12)
    R, k, seed = int(g['R']), g['rows'], int(g['seed'])
    nz = np.stack([philox.mag_normals(seed, run, 1000) for run in range(R)])
    mag = ins_np.mag_errors(g['ref_mag'], {'si': g['mag_si'], 'hi': g['mag_hi'], 'std': g['mag_std']}, nz)
    np.testing.assert_allclose(mag[:, k], g['mag'], rtol=0, atol=1e-12)
    z = [philox.gps_normals(seed, run, g['ref_gps'].shape[0]) for run in range(R)]
    gps = ins_np.gps_errors(g['ref_gps'], {'stdp': g['gps_stdp'], 'stdv': g['gps_stdv']}, rf,
                            np.stack([a for a, _ in z]), np.stack([b for _, b in z]))
    np.testing.assert_allclose(gps, g['gps'], rtol=1e-14, atol=1e-12)


def _t1_rates_cases():
    g = load_golden('t1_rates')
    for ci in range(int(g['count'])):
        c = {k[3:]: g[k] for k in g if k.startswith('c%d_' % ci)}
        for tag, rf, use_g, erot in (('extg', 0, True, False), ('wgs', 0, False, True), ('rf1', 1, False, True)):
            for plug in ('free', 'odo'):
                yield c, tag, rf, (c['ini'] if use_g else c['ini'][:9]), erot, plug


def test_t1_both_plugins_at_other_rates():
    """FreeIntegration.run of both plugins at 50 / 200 / 400 Hz on random band-limited records (reference-executed golden)."""
    for c, tag, rf, ini, erot, plug in _t1_rates_cases():
        k = c['rows']
        att, pos, vel = ins_np.free_integration(rf, float(c['fs']), c['gyro'][None], c['accel'][None] if plug == 'free' else None, ini,
                                                earth_rot=erot, odo=c['odo'][None] if plug == 'odo' else None)
        assert_traj_close(att[0][k], pos[0][k], vel[0][k], c['%s_%s_att' % (plug, tag)], c['%s_%s_pos' % (plug, tag)],
                          c['%s_%s_vel' % (plug, tag)], rtol=1e-11, what='%s %s %g Hz' % (plug, tag, float(c['fs'])))
