"""The drop-in surface on the GPU: the calling sequence of the reference's demo_free_integration.py, verbatim
except for the motion-profile path, reproduces the statistics the UNMODIFIED reference printed for the same
injected noise (golden t3_demo_rf1: Sim.run(4), algorithm=[odo, free], ref_frame=1)."""
import contextlib
import io
import math
import os

import numpy as np
import pytest

from conftest import load_golden, PKG, assert_traj_close

pytestmark = pytest.mark.gpu
D2R = math.pi / 180


def _demo_imu():
    return {'gyro_b': np.array([0.0, 0.0, 0.0]),
            'gyro_arw': np.array([0.25, 0.25, 0.25]) * 1.0,
            'gyro_b_stability': np.array([3.5, 3.5, 3.5]) * 1.0,
            'gyro_b_corr': np.array([100.0, 100.0, 100.0]),
            'accel_b': np.array([0.0e-3, 0.0e-3, 0.0e-3]),
            'accel_vrw': np.array([0.03119, 0.03009, 0.04779]) * 1.0,
            'accel_b_stability': np.array([4.29e-5, 5.72e-5, 8.02e-5]) * 1.0,
            'accel_b_corr': np.array([200.0, 200.0, 200.0]),
            'mag_std': np.array([0.2, 0.2, 0.2]) * 1.0}


def test_demo_free_integration_sequence(capsys):
    from gnss_ins_sim.sim import imu_model
    from gnss_ins_sim.sim import ins_sim
    from demo_algorithms import free_integration_odo
    from demo_algorithms import free_integration
    g = load_golden('t3_demo_rf1')
    fs = 100.0
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy=_demo_imu(), axis=6, gps=False, odo=True, odo_opt={'scale': 0.999, 'stdv': 0.1})
    ini_pos_vel_att = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini_pos_vel_att[0] = ini_pos_vel_att[0] * D2R
    ini_pos_vel_att[1] = ini_pos_vel_att[1] * D2R
    ini_pos_vel_att[6:9] = ini_pos_vel_att[6:9] * D2R
    algo1 = free_integration_odo.FreeIntegration(ini_pos_vel_att)
    algo2 = free_integration.FreeIntegration(ini_pos_vel_att)
    sim = ins_sim.Sim([fs, 0.0, 0.0], csv, ref_frame=1, imu=imu, mode=None, env=None, algorithm=[algo1, algo2],
                      seed=int(g['seed']))
    sim.run(int(g['R']))
    avail = sim.results(err_stats_start=-1, gen_kml=True)
    text = capsys.readouterr().out
    assert 'Simulation runs: 4' in text and 'The following are error statistics.' in text
    assert 'Simulation run algo0:' in text and 'Simulation run algo1:' in text
    for name in ('att_euler', 'pos', 'vel', 'accel', 'gyro', 'odo', 'ref_pos', 'att_quat', 'ref_att_quat'):
        assert name in avail, name
    # statistics == what the reference computed on the same noise
    for dn in ('att_euler', 'pos', 'vel'):
        st = sim.err_stats[dn]
        for s in ('max', 'avg', 'std'):
            for grp in ('algo0', 'algo1'):
                np.testing.assert_allclose(st[s][grp], g['stat_%s_%s_%s' % (dn, s, grp)], rtol=1e-6, atol=1e-11)
    # data views: per-run arrays exactly like dmgr.<series>.data[key] of the reference
    k = g['rows']
    d = sim.dmgr
    assert len(d.accel.data) == 4 and list(d.accel.data.keys()) == [0, 1, 2, 3]
    for r in range(4):
        np.testing.assert_allclose(d.accel.data[r][k], g['accel'][r], rtol=0, atol=1e-12)
        np.testing.assert_allclose(d.gyro.data[r][k], g['gyro'][r], rtol=0, atol=1e-14)
        np.testing.assert_allclose(d.odo.data[r][k], g['odo'][r], rtol=0, atol=1e-12)
        assert_traj_close(d.att_euler.data['algo0_%d' % r][k], d.pos.data['algo0_%d' % r][k], d.vel.data['algo0_%d' % r][k],
                          g['odo_att'][r], g['odo_pos'][r], g['odo_vel'][r], rtol=1e-9, what='odo')
        assert_traj_close(d.att_euler.data['algo1_%d' % r][k], d.pos.data['algo1_%d' % r][k], d.vel.data['algo1_%d' % r][k],
                          g['fi_att'][r], g['fi_pos'][r], g['fi_vel'][r], rtol=1e-9, what='fi')
    assert d.att_quat.data['algo1_2'].shape == (1000, 4)
    assert 'algo1_4' not in d.pos.data and 7 not in d.accel.data
    assert algo1.run_times == 4 and algo2.run_times == 4
    data = sim.get_data(['ref_frame', 'fs', 'gyro'])
    assert data[0] == 1 and data[1] == fs and data[2][3].shape == (1000, 3)


@pytest.mark.parametrize('name,accuracy,rf,odo_opt,algo_tags,keep', [
    ('t3_vib_random_rf1', 'mid-accuracy', 1, None, ['fi'], True),
    ('t3_vib_sin_rf0', 'low-accuracy', 0, {'scale': 0.999, 'stdv': 0.1}, ['fi', 'odo'], True),
    ('t3_vib_sin_rf0', 'low-accuracy', 0, {'scale': 0.999, 'stdv': 0.1}, ['fi', 'odo'], False),
    ('t3_vib_mixed_rf1', _demo_imu(), 1, None, ['fi'], False),
    # env as an (n, 4) PSD array (ins_sim.py:686-697): the accelerometer's interpolated to the series' grid, the gyroscope's given on it
    ('t3_vib_psd_rf1', 'mid-accuracy', 1, None, ['fi'], True),
    ('t3_vib_psd_rf1', 'mid-accuracy', 1, None, ['fi'], False)])
def test_sim_with_a_vibration_environment(name, accuracy, rf, odo_opt, algo_tags, keep, capsys):
    """Sim(env={'acc': ..., 'gyro': ...}) (ins_sim.py:108-124, 482-495): the env STRINGS the golden recipe gave the unmodified
    reference, through the drop-in Sim -- sensor series per sample, end-point and process statistics as the reference computed
    them on the same injected noise.  keep=False: nothing materialised, the statistics come from the kernels' accumulators
    (the vibration variants of the online-statistics kernels)."""
    from gnss_ins_sim.sim import imu_model
    from gnss_ins_sim.sim import ins_sim
    from demo_algorithms import free_integration_odo
    from demo_algorithms import free_integration
    g = load_golden(name)
    env = {k: (g['env_' + k].copy() if g['env_' + k].ndim == 2 else str(g['env_' + k])) for k in ('acc', 'gyro') if 'env_' + k in g}
    assert env
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy=accuracy, axis=6, gps=False, odo=odo_opt is not None, odo_opt=odo_opt)
    mods = {'fi': free_integration, 'odo': free_integration_odo}
    algos = [mods[t].FreeIntegration(g['ini'].copy()) for t in algo_tags]
    R = int(g['R'])
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=rf, imu=imu, mode=None, env=env, algorithm=algos, seed=int(g['seed']),
                      keep_trajectories=keep, keep_runs=0 if keep else 2, stats_start=2.0)
    sim.run(R)
    for sensor, e in env.items():           # a PSD given on the series' grid is left as the reference leaves the caller's array: halved
        if isinstance(e, np.ndarray):       # once per run (time_series_from_psd.py:44-49 works in place when it need not interpolate)
            np.testing.assert_allclose(e, g['env_%s_after' % sensor], rtol=1e-15, atol=0)
    k = g['rows']
    d = sim.dmgr
    for r in range(R if keep else 2):
        np.testing.assert_allclose(d.accel.data[r][k], g['accel'][r], rtol=0, atol=1e-12)
        np.testing.assert_allclose(d.gyro.data[r][k], g['gyro'][r], rtol=0, atol=1e-14)
        for ai, t in enumerate(algo_tags):
            key = 'algo%d_%d' % (ai, r) if len(algo_tags) > 1 else 'algo0_%d' % r
            assert_traj_close(d.att_euler.data[key][k], d.pos.data[key][k], d.vel.data[key][k], g[t + '_att'][r], g[t + '_pos'][r],
                              g[t + '_vel'][r], rtol=1e-9, what=name + t)
    sim.results(err_stats_start=-1)
    for dn in ('att_euler', 'pos', 'vel'):
        st = sim.err_stats[dn]
        for s in ('max', 'avg', 'std'):
            for ai in range(len(algo_tags)):
                got = st[s]['algo%d' % ai] if hasattr(st[s], 'keys') else st[s]
                np.testing.assert_allclose(got, g['stat_%s_%s_algo%d' % (dn, s, ai)], rtol=1e-6, atol=1e-11)
    # process statistics from t = 2 s (internal units in the golden)
    sim.results(err_stats_start=2.0)
    keys = ['algo%d_%d' % (ai, r) for ai in range(len(algo_tags)) for r in range(R)]
    r2d = 180.0 / math.pi
    unit = {'att_euler': np.full(3, r2d), 'pos': np.array([r2d, r2d, 1.0]) if rf == 0 else np.ones(3), 'vel': np.ones(3)}
    for dn in ('att_euler', 'pos', 'vel'):
        st = sim.err_stats[dn]
        for s in ('max', 'avg', 'std'):
            got = np.stack([np.asarray(st[s][kk]) for kk in keys])
            np.testing.assert_allclose(got, g['proc_%s_%s' % (dn, s)] * unit[dn], rtol=2e-6, atol=1e-9)
    capsys.readouterr()


def test_plugin_run_given_data_like_openimu_demo():
    """demo_free_integration_openimu.py recipe: plugin.run() on logged data, ref_frame 0, external gravity."""
    from demo_algorithms import free_integration
    g = load_golden('t1_fixture_bosch')
    algo = free_integration.FreeIntegration(g['ini'], earth_rot=False)
    algo.run([0, 100.0, g['gyro'], g['accel']])
    att, pos, vel = algo.get_results()
    k = g['rows']
    assert_traj_close(att[k], pos[k], vel[k], g['att_extg'], g['pos_extg'], g['vel_extg'], rtol=1e-10)


def test_stats_only_large_run_and_seed_convention():
    """65 536 runs, keep_trajectories=False: statistics only; np.random.seed() makes it repeatable."""
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= D2R
    ini[6:9] *= D2R
    out = []
    for _ in range(2):
        np.random.seed(5)
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
        sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=free_integration.FreeIntegration(ini),
                          keep_trajectories=False)
        sim.run(65536)
        sim.results(err_stats_start=-1)
        out.append(sim.err_stats)
    np.testing.assert_array_equal(out[0]['vel']['std'], out[1]['vel']['std'])
    att_std = out[0]['att_euler']['std']
    assert np.all(np.abs(att_std - 0.01318) < 0.01318 * 0.03), att_std          # ARW*sqrt(T), deg
    assert 'accel' not in sim.dmgr.available


def test_saved_csv_files_match_the_reference(tmp_path):
    """Sim.results(data_dir): same file names, same header lines ('legend (unit)', sim_data.py:117-165) and the same numbers
    as the files the unmodified reference wrote for this case (two runs, two algorithms, rf 0, GPS + odometer; the
    reference fed the engine's normals).  Angles are in degrees in the files, so differences modulo 360 count."""
    from conftest import load_golden
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration, free_integration_odo
    g = load_golden('csv_files_rf0')
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=True, odo=True,
                        odo_opt={'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])})
    ini = g['ini']
    algos = [free_integration.FreeIntegration(ini.copy()), free_integration_odo.FreeIntegration(ini.copy())]
    sim = ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=0, imu=imu, mode=None, env=None, algorithm=algos, seed=int(g['seed']))
    sim.run(2)
    out = str(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(out, err_stats_start=-1)
    ours = sorted(f for f in os.listdir(out) if f.endswith('.csv'))
    want = [str(x) for x in g['names']]
    assert ours == want
    for f, head, shape in zip(want, g['headers'], g['shapes']):
        with open(os.path.join(out, f)) as fh:
            assert fh.readline().rstrip('\n') == str(head), f
        a = np.atleast_1d(np.genfromtxt(os.path.join(out, f), delimiter=',', skip_header=1))
        if a.ndim == 1:
            a = a[:, None]
        assert a.shape == tuple(shape), f
        got, ref = a[g['rows_' + f]], g['data_' + f]
        if '(deg)' in str(head) and 'pos' not in f and 'gps' not in f:
            d = np.mod(got - ref + 180.0, 360.0) - 180.0
            assert np.max(np.abs(d)) < 1e-7, f
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9, err_msg=f)


def _tokens(line):
    import re
    return [t for t in re.split(r'([\[\]\s,])', line) if t.strip()]


@pytest.mark.parametrize('tag,start,opt', [('end', -1, ''), ('process', 0, ''), ('end_ned', -1, 'ned')])
def test_printed_summary_is_the_reference_text(tag, start, opt):
    """Sim.results() prints what the unmodified reference printed for the same injected noise (Sim.__summary,
    ins_sim.py:339-413): the same lines in the same order, every word equal, every number within 2e-6 relative
    (the text carries 8-9 digits) -- for the end-point statistics, the process statistics of the reference's default
    err_stats_start = 0 and the end-point statistics in NED."""
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration, free_integration_odo
    g = load_golden('summary_text_rf0')
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=True, odo=True,
                        odo_opt={'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])})
    ini = g['ini']
    algos = [free_integration.FreeIntegration(ini.copy()), free_integration_odo.FreeIntegration(ini.copy())]
    sim = ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=0, imu=imu, mode=None, env=None, algorithm=algos, seed=int(g['seed']))
    sim.run(2)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sim.results(err_stats_start=start, extra_opt=opt)
    ours = [l for l in buf.getvalue().split('\n')]
    want = [l for l in str(g['text_' + tag]).split('\n')]
    assert len(ours) == len(want), '\n'.join(ours)
    for lo, lw in zip(ours, want):
        to, tw = _tokens(lo), _tokens(lw)
        assert len(to) == len(tw), (lo, lw)
        for a, b in zip(to, tw):
            try:
                fb = float(b)
            except ValueError:
                assert a == b, (lo, lw)
                continue
            fa = float(a)
            assert abs(fa - fb) <= 2e-6 * abs(fb) + 1e-12, (lo, lw)


def test_kept_runs_and_statistics_launch_start_on_the_dies_that_fit(capsys):
    """Sim(keep_runs=K) on a statistics-only batch runs the kept runs as one workgroup on a sibling context next to the statistics
    launch over the others.  The dispatcher deals a launch's workgroups to the XCDs in turn from a die that belongs to the stream's
    hardware queue (ginsim_stream_first_xcc): W statistics workgroups leave their spare slot on die (first + W) mod 8, and the
    kept runs' workgroup must land THERE (C3: 0.93 s for the pair, 1.20 s on any other die).  The pair the Sim settles on obeys
    that, also after another library of the process has made streams of its own (here: a PSD vibration's FFT plans)."""
    import ginsim
    from ginsim import workloads
    from gnss_ins_sim.sim import ins_sim
    ctx = ginsim.default_context()
    assert 0 <= ctx.first_xcc() <= 7 and ctx.first_xcc() == ctx.first_xcc()
    f = np.array([0.0, 8.0, 20.0, 50.0])
    v = {'type': 'psd', 'freq': f, 'x': np.full(4, 1e-3), 'y': np.full(4, 1e-3), 'z': np.full(4, 2e-3)}
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=256, seed=3, keep_sensors=True, vib_accel=v).run().release()
    sim = ins_sim.Sim.__new__(ins_sim.Sim)
    sim._side_ctx = None
    for w in (1023, 1023, 511, 5):
        blk, rest = sim._block_and_rest_contexts(ctx, w)
        assert (blk is ctx) != (rest is ctx) and blk.device == rest.device == ctx.device
        mine = ctx.first_xcc()
        xs = sorted({c.first_xcc() for c in ins_sim.Sim._SIBLINGS[ctx.device]['spare'] + [sim._side_ctx]})
        fits = blk.first_xcc() == (rest.first_xcc() + w) % 8
        possible = any(x == (mine + w) % 8 or mine == (x + w) % 8 for x in xs)       # one of the two is always ctx
        assert fits or not possible, (w, blk.first_xcc(), rest.first_xcc(), mine, xs)
        assert fits or w == 5, (w, mine, xs)    # four hardware queues on consecutive dies: +1 / -1 always exist, +5 need not
        assert blk.first_xcc() != rest.first_xcc()      # never two streams of one hardware queue: they would run one after the other
    capsys.readouterr()
