"""Process-error statistics (err_stats_start >= 0) and NED position error (extra_opt='ned'):
oracle vs the reference goldens (CPU) and device vs both (GPU)."""
import numpy as np
import pytest

from conftest import load_golden

CASES = ['t3_demo_rf1', 't3_mid_rf0', 't3_white_gps_rf0', 't3_low_rf1', 't3_high_odo_rf0', 't3_drive200_rf0']


def _errs(g):
    acc = {k[6:]: g[k] for k in g if k.startswith('accel_') and k != 'accel'}
    gyr = {k[5:]: g[k] for k in g if k.startswith('gyro_') and k != 'gyro'}
    return acc, gyr


def _algos(g):
    # run keys are 'algo<i>_<r>' in the order the case listed its algorithms
    order = {'t3_demo_rf1': ['odo', 'free'], 't3_mid_rf0': ['free'], 't3_white_gps_rf0': ['free', 'odo'],
             't3_low_rf1': ['free'], 't3_high_odo_rf0': ['odo', 'free'], 't3_drive200_rf0': ['free', 'odo']}
    return order


def _want(g, ai, R):
    rows = slice(ai * R, (ai + 1) * R)
    w = np.empty((R, 3, 9))
    for si, s in enumerate(('max', 'avg', 'std')):
        w[:, si, 0:3] = g['proc_att_euler_' + s][rows]
        w[:, si, 3:6] = g['proc_pos_' + s][rows]
        w[:, si, 6:9] = g['proc_vel_' + s][rows]
    return w


@pytest.mark.parametrize('name', CASES)
def test_oracle_process_stats_match_reference(name):
    from oracle import ins_np
    g = load_golden(name)
    R, fs, rf = int(g['R']), float(g['fs']), int(g['ref_frame'])
    acc, gyr = _errs(g)
    runs = np.arange(R)
    accel, gyro = ins_np.mc_sensors(int(g['seed']), runs, fs, g['ref_accel'], g['ref_gyro'], acc, gyr)
    odo = None
    if 'odo' in g:
        odo = ins_np.mc_odo(int(g['seed']), runs, g['ref_odo'], {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])})
    j0 = int(round(float(g['proc_start_s']) * fs))
    for ai, a in enumerate(_algos(g)[name]):
        att, pos, vel = ins_np.free_integration(rf, fs, gyro, accel, g['ini'], odo=odo if a == 'odo' else None)
        st = ins_np.process_error_stats(att, pos, vel, g['ref_att'], g['ref_pos'], g['ref_vel'], j0)
        np.testing.assert_allclose(st, _want(g, ai, R), rtol=1e-7, atol=1e-11)
        if rf == 0:
            ned = ins_np.process_error_stats(att, pos, vel, g['ref_att'], g['ref_pos'], g['ref_vel'], j0, pos_ned=True)
            for si, s in enumerate(('max', 'avg', 'std')):
                np.testing.assert_allclose(ned[:, si, 3:6], g['ned_proc_' + s][ai * R:(ai + 1) * R], rtol=1e-6, atol=1e-8)
            e_end = ins_np.lla_error_ned(pos[:, -1], np.broadcast_to(g['ref_pos'][-1], (R, 3)))
            grp = 'algo%d' % ai
            np.testing.assert_allclose(np.abs(e_end).max(0), g['ned_end_max_' + grp], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(e_end.std(0), g['ned_end_std_' + grp], rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_device_process_stats_and_ned_match_reference(name):
    import ginsim
    g = load_golden(name)
    ctx = ginsim.default_context()
    R, fs, rf = int(g['R']), float(g['fs']), int(g['ref_frame'])
    acc, gyr = _errs(g)
    truth = {'ref_accel': g['ref_accel'], 'ref_gyro': g['ref_gyro'], 'ref_pos': g['ref_pos'], 'ref_vel': g['ref_vel'],
             'ref_att': g['ref_att']}
    odo_err = None
    if 'odo' in g:
        truth['ref_odo'] = g['ref_odo']
        odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])}
    algos = _algos(g)[name]
    job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, g['ini'], runs=R, algos=tuple(algos), odo_err=odo_err,
                               seed=int(g['seed']), keep_traj=True, end_pos_ned=(rf == 0)).run()
    j0 = int(round(float(g['proc_start_s']) * fs))
    for ai, a in enumerate(algos):
        st = job.process_stats(a, j0)
        np.testing.assert_allclose(st, _want(g, ai, R), rtol=1e-6, atol=1e-10)
        if rf == 0:
            ned = job.process_stats(a, j0, pos_ned=True)
            for si, s in enumerate(('max', 'avg', 'std')):
                np.testing.assert_allclose(ned[:, si, 3:6], g['ned_proc_' + s][ai * R:(ai + 1) * R], rtol=1e-6, atol=2e-8)
            end = job.stats(a)              # end-point position error already in NED metres
            grp = 'algo%d' % ai
            np.testing.assert_allclose(end.maxabs[3:6], g['ned_end_max_' + grp], rtol=1e-6, atol=2e-8)
            np.testing.assert_allclose(end.std[3:6], g['ned_end_std_' + grp], rtol=1e-5, atol=2e-8)
    job.release()


@pytest.mark.gpu
def test_sim_results_process_mode_and_ned():
    """Sim.results(err_stats_start=2.0) and extra_opt='ned' through the drop-in == the reference's numbers."""
    import os
    from conftest import PKG
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    g = load_golden('t3_mid_rf0')
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=0, imu=imu, algorithm=free_integration.FreeIntegration(g['ini']),
                      seed=int(g['seed']))
    sim.run(int(g['R']))
    sim.results(err_stats_start=2.0, extra_opt='ned')
    st = sim.err_stats
    assert sorted(st['vel']['max'].keys()) == ['algo0_0', 'algo0_1', 'algo0_2', 'algo0_3']
    r2d = 180 / np.pi
    for r in range(4):
        np.testing.assert_allclose(st['att_euler']['std']['algo0_%d' % r], g['proc_att_euler_std'][r] * r2d, rtol=1e-6)
        np.testing.assert_allclose(st['vel']['avg']['algo0_%d' % r], g['proc_vel_avg'][r], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(st['pos']['max']['algo0_%d' % r], g['ned_proc_max'][r], rtol=1e-6, atol=2e-8)
    assert st['pos']['units'] == "['m', 'm', 'm']"
    end = sim.dmgr.get_error_stats('pos', err_stats_start=-1, extra_opt='ned')
    np.testing.assert_allclose(end['max'], g['ned_end_max_algo0'], rtol=1e-6, atol=2e-8)


# ------------------------------------------------------------------------------------------------------------------
# Statistics WITHOUT trajectories (SURVEY 8(f) rank 2 "at scale"): the fused kernel accumulates the
# process-error statistics online and writes a second, NED end-point record.
@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_online_process_stats_match_reference_and_the_kept_path(name):
    import ginsim
    g = load_golden(name)
    ctx = ginsim.default_context()
    R, fs, rf = int(g['R']), float(g['fs']), int(g['ref_frame'])
    acc, gyr = _errs(g)
    truth = {'ref_accel': g['ref_accel'], 'ref_gyro': g['ref_gyro'], 'ref_pos': g['ref_pos'], 'ref_vel': g['ref_vel'],
             'ref_att': g['ref_att']}
    odo_err = None
    if 'odo' in g:
        truth['ref_odo'] = g['ref_odo']
        odo_err = {'scale': float(g['odo_scale']), 'stdv': float(g['odo_stdv'])}
    j0 = int(round(float(g['proc_start_s']) * fs))
    for ai, a in enumerate(_algos(g)[name]):
        kept = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, g['ini'], runs=R, algos=(a,), odo_err=odo_err,
                                    seed=int(g['seed']), keep_traj=True).run()
        for first in (j0, 0):
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, g['ini'], runs=R, algos=(a,), odo_err=odo_err,
                                       seed=int(g['seed']), proc_first=first, end_ned=(rf == 0)).run()
            st = job.process_stats_online(a)
            if first == j0:
                np.testing.assert_allclose(st, _want(g, ai, R), rtol=1e-6, atol=1e-10)          # the reference's numbers
            # the same recurrence over the same samples as the kernel that reads kept trajectories: equal to the bit,
            # except that one splits the window into four segments and Chan-merges them
            np.testing.assert_allclose(st, kept.process_stats(a, first), rtol=1e-9, atol=1e-14)
            np.testing.assert_array_equal(job.end_errors(a), kept.end_errors(a))                 # same trajectories
            if rf == 0:
                grp = 'algo%d' % ai
                end = job.stats(a, ned=True)
                np.testing.assert_allclose(end.maxabs[3:6], g['ned_end_max_' + grp], rtol=1e-6, atol=2e-8)
                np.testing.assert_allclose(end.std[3:6], g['ned_end_std_' + grp], rtol=1e-5, atol=2e-8)
                np.testing.assert_array_equal(end.maxabs[[0, 1, 2, 6, 7, 8]], job.stats(a).maxabs[[0, 1, 2, 6, 7, 8]])
            job.release()
        if rf == 0:
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, g['ini'], runs=R, algos=(a,), odo_err=odo_err,
                                       seed=int(g['seed']), proc_first=j0, proc_ned=True).run()
            ned = job.process_stats_online(a)
            for si, s in enumerate(('max', 'avg', 'std')):
                np.testing.assert_allclose(ned[:, si, 3:6], g['ned_proc_' + s][ai * R:(ai + 1) * R], rtol=1e-6, atol=2e-8)
            job.release()
        kept.release()


@pytest.mark.gpu
def test_online_process_stats_argument_checks():
    import ginsim
    from ginsim import workloads
    ctx = ginsim.default_context()
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    with pytest.raises(ValueError, match='one algorithm'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4, algos=('free', 'odo'),
                             odo_err={'scale': 1.0, 'stdv': 0.1}, proc_first=0)
    with pytest.raises(ValueError, match='ref_frame 0'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4, proc_first=0, proc_ned=True)
    with pytest.raises(ValueError, match='sample index'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4, proc_first=1000)


@pytest.mark.gpu
def test_sim_stats_only_results_with_reference_defaults_and_keep_runs(tmp_path):
    """Sim in stats-only mode (what a 262 144-run Sim is): results() with the reference's defaults (process statistics from
    t = 0), another window, extra_opt='ned', and keep_runs=K materialising sensors + GPS + outputs of the first K runs --
    all equal to the same Sim with everything kept."""
    import contextlib
    import io
    import os
    from conftest import PKG
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration, free_integration_odo
    g = load_golden('t3_mid_rf0')
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    R, K = 300, 3

    def make(**kw):
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=True, odo=True, odo_opt={'scale': 0.999, 'stdv': 0.1})
        algos = [free_integration.FreeIntegration(g['ini']), free_integration_odo.FreeIntegration(g['ini'])]
        return ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=0, imu=imu, algorithm=algos, seed=99, **kw)

    full = make(keep_trajectories=True)
    lean = make(keep_trajectories=False, keep_runs=K)
    for sim in (full, lean):
        sim.run(R)
    assert lean.kept is False and set(lean.dmgr.accel.data.keys()) == set(range(K)) and len(full.dmgr.accel.data) == R
    for r in range(K):
        np.testing.assert_array_equal(lean.dmgr.gyro.data[r], full.dmgr.gyro.data[r])
        np.testing.assert_array_equal(lean.dmgr.gps.data[r], full.dmgr.gps.data[r])
        np.testing.assert_array_equal(lean.dmgr.odo.data[r], full.dmgr.odo.data[r])
        for a in ('algo0', 'algo1'):
            np.testing.assert_array_equal(lean.dmgr.pos.data['%s_%d' % (a, r)], full.dmgr.pos.data['%s_%d' % (a, r)])
    assert 'algo0_%d' % K not in lean.dmgr.pos.data

    def stats(sim, **kw):
        with contextlib.redirect_stdout(io.StringIO()) as out:
            sim.results(**kw)
        return sim.err_stats, out.getvalue()

    for kw in (dict(), dict(err_stats_start=2.0), dict(err_stats_start=2.0, extra_opt='ned'), dict(err_stats_start=-1),
               dict(err_stats_start=-1, extra_opt='ned')):
        a, text_a = stats(full, **kw)
        b, text_b = stats(lean, **kw)
        for name in ('att_euler', 'pos', 'vel'):
            assert a[name]['units'] == b[name]['units']
            # NED metres are differences of ECEF coordinates (6.4e6 m, ulp 1e-9 m): the two kernels that evaluate them
            # are compiled with different FMA contraction and agree to 2e-10 m
            atol = 2e-9 if (name == 'pos' and kw.get('extra_opt') == 'ned') else 1e-13
            for s in ('max', 'avg', 'std'):
                if kw.get('err_stats_start', 0) == -1:
                    assert set(a[name][s].keys()) == {'algo0', 'algo1'}
                    for grp in ('algo0', 'algo1'):
                        np.testing.assert_allclose(b[name][s][grp], a[name][s][grp], rtol=1e-9, atol=atol)
                else:
                    assert len(b[name][s]) == 2 * R
                    for key in ('algo0_0', 'algo1_7', 'algo0_%d' % (R - 1), 'algo1_%d' % (R - 1)):
                        np.testing.assert_allclose(b[name][s][key], a[name][s][key], rtol=1e-9, atol=atol)
        if kw.get('err_stats_start', 0) != -1:          # every run is listed, as the reference does (ins_sim.py:387-392)
            assert 'Simulation run algo1_%d:' % (R - 1) in text_b and 'Simulation run algo0_150:' in text_b and 'more runs' not in text_b
    _, short = stats(lean, max_summary_runs=5)          # an explicit limit truncates by (algorithm, run NUMBER)
    short = short[short.rindex('Sample frequency of IMU'):]      # Sim.sum accumulates the summaries of every results() call, as the reference's does
    assert '... %d more runs' % (2 * R - 5) in short and 'Simulation run algo0_4:' in short and 'Simulation run algo0_10:' not in short
    with contextlib.redirect_stdout(io.StringIO()):
        lean.results(str(tmp_path), err_stats_start=-1)
    files = set(os.listdir(tmp_path))
    assert {'accel-0.csv', 'gps-2.csv', 'pos-algo1_2.csv', 'ref_gps.csv'} <= files and 'accel-%d.csv' % K not in files
    # single precision, statistics only: the default results() now gives the per-run process statistics too (the runs are
    # re-integrated block by block with float trajectories kept, tests/test_gpu_fp32.py), no fall-back notice
    f32 = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False),
                      algorithm=free_integration.FreeIntegration(g['ini']), seed=99, precision='f32', keep_trajectories=False)
    f32.run(64)
    st, text = stats(f32)
    assert 'end-point statistics' not in text and len(st['vel']['std']) == 64 and 'Simulation run algo0_63:' in text


@pytest.mark.gpu
@pytest.mark.parametrize('rf,algo', [(1, 'free'), (1, 'odo'), (0, 'odo'), (0, 'free')])
def test_online_statistics_floor_for_a_constant_error(rf, algo):
    """ADVICE r03 / VERDICT r04 item 8, r05 remark P1: an error that is nearly constant over the window -- here an ideal IMU (no
    noise at all) started 1e-3 rad / 0.5 m/s away from the truth.  The online accumulator keeps its sums about an error close to
    the mean (Proc in csrc/mc_kernel.hip: the run's first in-window error where the kernel has nine registers for it, the
    launch's error at sample 0 in the ref_frame 0 free-integration kernel -- C3's -- and the vibration variants), so the std is
    the TRUE small value, as the kept-trajectory path (Welford) and the reference's np.std (ins_data_manager.py:761-795) give
    it; the raw sums round 5 still ran in C3's kernel were off by ~1.5e-8 |mean| here."""
    import ginsim
    g = load_golden('t2_turn_rf%d' % rf)
    r = ginsim.pathgen(g['ini_pva'], g['motion_def'], 100.0, 0.0, g['mobility'], rf)
    truth = {'ref_accel': r['imu'][:, 1:4], 'ref_gyro': r['imu'][:, 4:7], 'ref_pos': r['nav'][:, 1:4], 'ref_vel': r['nav'][:, 4:7],
             'ref_att': r['nav'][:, 7:10], 'ref_odo': r['odo'][:, 2]}
    ideal = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, np.inf), 'arw': np.zeros(3), 'vrw': np.zeros(3)}
    ini = np.array(g['ini_pva'], dtype=np.float64)
    ini[6] += 1e-3
    ini[3] += 0.5
    ctx = ginsim.default_context()
    kw = dict(runs=64, seed=1, algos=(algo,), odo_err={'scale': 1.0, 'stdv': 0.0})
    for first in (0, 117):
        online = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, ideal, ideal, ini, proc_first=first, **kw).run()
        kept = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, ideal, ideal, ini, keep_traj=True, **kw).run()
        a, b = online.process_stats_online(algo), kept.process_stats(algo, first)
        np.testing.assert_allclose(a[:, 0], b[:, 0], rtol=1e-12, atol=1e-15)                        # max |e|
        np.testing.assert_allclose(a[:, 1], b[:, 1], rtol=1e-11, atol=1e-15)                        # mean
        assert np.all(b[:, 2, 0] < 1e-4 * np.abs(b[:, 1, 0]))           # the case really is "constant error": yaw std << |yaw mean|
        np.testing.assert_allclose(a[:, 2], b[:, 2], rtol=1e-7, atol=1e-13 * np.abs(b[:, 1]).max() + 1e-18)
        assert np.any(b[:, 2] < 1e-4 * np.abs(b[:, 1]))             # (the raw form is off by ~1.5e-8 |mean| here: >> 1e-7 std)
        online.release()
        kept.release()


@pytest.mark.gpu
@pytest.mark.parametrize('ned', [False, True])
def test_runs_that_start_on_the_truth_take_plain_sums_and_lose_nothing(ned):
    """ginsim_mc_params.proc_plain_sums: the ref_frame 0 free-integration statistics kernels shift their sums about ONE error per
    launch (the first run's at sample 0); MonteCarloJob sees that the initial state IS the truth's first sample, states it, and the
    launch runs the form without the nine subtractions (C3: 2.3 %).  Forced back to the shifted form the numbers are the same: the
    maxima to the bit (they never see the shift), means and deviations to rounding."""
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 0)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    ctx = ginsim.default_context()
    kw = dict(runs=512, seed=5, algos=('free',), proc_first=0, proc_ned=ned)
    plain = ginsim.MonteCarloJob(ctx, 100.0, 0, truth, acc, gyr, ini, **kw)
    assert plain.params.proc_plain_sums == 1 and plain.kernel_name() == 'ginsim::mc_kernel<0, 1, false, %s, false>' % ('true, 4' if ned else 'false, 3')
    shifted = ginsim.MonteCarloJob(ctx, 100.0, 0, truth, acc, gyr, ini, **kw)
    shifted.params.proc_plain_sums = 0
    assert shifted.kernel_name() == 'ginsim::mc_kernel<0, 1, false, %s, false>' % ('true, 2' if ned else 'false, 1')
    a, b = plain.run().process_stats_online('free'), shifted.run().process_stats_online('free')
    np.testing.assert_array_equal(a[:, 0], b[:, 0])
    np.testing.assert_allclose(a[:, 1], b[:, 1], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(a[:, 2], b[:, 2], rtol=1e-10, atol=1e-18)
    np.testing.assert_array_equal(plain.end_errors('free'), shifted.end_errors('free'))
    off = np.array(ini, dtype=np.float64)
    off[6] += 1e-3                                  # off the truth: the job does not state it
    assert ginsim.MonteCarloJob(ctx, 100.0, 0, truth, acc, gyr, off, **kw).params.proc_plain_sums == 0
    bad = ginsim.MonteCarloJob(ctx, 100.0, 0, truth, acc, gyr, ini, **kw)
    bad.params.proc_plain_sums = 2
    with pytest.raises(ValueError, match='proc_plain_sums'):
        bad.run()
    for j in (plain, shifted, bad):
        j.release()


@pytest.mark.gpu
@pytest.mark.parametrize('rf,algo', [(0, 'free'), (0, 'odo'), (1, 'free'), (1, 'odo')])
def test_simple_model_statistics_kernel_is_bit_identical_to_the_general_one(monkeypatch, rf, algo):
    """Round 6: launches with online process statistics take the SIMPLE-sensor-model instantiation
    (mc_kernel<RF, ALGO, false, false, 1>: no white-drift axis, no constant bias -- every standard IMU grade) where round 5 always
    took the general one; C3 (rf 0, free) 0.99 -> 0.92 s.  x + 0.0 and the compiled-out selects change no bit: per-run statistics
    and both end-point records equal the general kernel's ($GINSIM_PS_GENERAL forces it), which the goldens pin."""
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    ctx = ginsim.default_context()
    kw = dict(runs=700, seed=77, algos=(algo,), odo_err={'scale': 0.999, 'stdv': 0.1}, proc_first=13, end_ned=(rf == 0))
    got = []
    for general in (False, True):
        if general:
            monkeypatch.setenv('GINSIM_PS_GENERAL', '1')
        job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, **kw).run()
        form = 3 if (rf, algo) == (0, 'free') else 1        # the runs start on the truth: plain sums where the shift is per launch
        assert job.kernel_name() == 'ginsim::mc_kernel<%d, %d, false, %s, %d, false>' % (rf, 1 if algo == 'free' else 2, 'true' if general else 'false', form)
        got.append([job.process_stats_online(algo).copy(), job.end_errors(algo)] + ([job.end_errors(algo, ned=True)] if rf == 0 else []))
        job.release()
    for a, b in zip(*got):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('stats_start', [0, 2.0, -1])
def test_kept_runs_riding_along_as_the_first_workgroup_change_nothing(monkeypatch, stats_start):
    """Statistics-only Sim with keep_runs: the kept runs are integrated as the first 256-run workgroup of the batch on a sibling
    context while the other runs are integrated statistics-only (ins_sim._BlockAndRest), instead of in a small launch of their own in
    front of a launch over all runs.  Same kept series (bit for bit), same per-run process statistics (bit for bit: the same
    kernel accumulates them for the same global run ids), same end-point statistics up to the association of the Chan merge."""
    import contextlib
    import io
    import os
    from conftest import PKG
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration, free_integration_odo
    g = load_golden('t3_mid_rf0')
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    R, K = 700, 4

    def run(ride):
        monkeypatch.setattr(ins_sim, 'KEPT_BLOCK', 256 if ride else 10 ** 9)
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=True, odo=True, odo_opt={'scale': 0.999, 'stdv': 0.1})
        algos = [free_integration.FreeIntegration(g['ini']), free_integration_odo.FreeIntegration(g['ini'])]
        sim = ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=0, imu=imu, algorithm=algos, seed=17, keep_trajectories=False, keep_runs=K,
                          stats_start=stats_start, device=0)
        sim.run(R)
        assert sim.mc.kept_block is ride
        with contextlib.redirect_stdout(io.StringIO()):
            sim.results(err_stats_start=stats_start)
        a = sim.err_stats
        with contextlib.redirect_stdout(io.StringIO()):
            sim.results(err_stats_start=stats_start, extra_opt='ned')
        return sim, a, sim.err_stats

    s1, a1, n1 = run(True)
    s0, a0, n0 = run(False)
    assert set(s1.dmgr.accel.data.keys()) == set(range(K))
    for r in range(K):
        for nm in ('accel', 'gyro', 'odo', 'gps'):
            np.testing.assert_array_equal(getattr(s1.dmgr, nm).data[r], getattr(s0.dmgr, nm).data[r])
        for a in ('algo0', 'algo1'):
            for nm in ('pos', 'vel', 'att_euler'):
                np.testing.assert_array_equal(getattr(s1.dmgr, nm).data['%s_%d' % (a, r)], getattr(s0.dmgr, nm).data['%s_%d' % (a, r)])
    for x, y in ((a1, a0), (n1, n0)):
        for name in ('att_euler', 'pos', 'vel'):
            for st in ('max', 'avg', 'std'):
                assert sorted(x[name][st].keys()) == sorted(y[name][st].keys())
                for key in x[name][st].keys():
                    if stats_start == -1:       # end-point statistics over all runs: two partial records merged
                        np.testing.assert_allclose(x[name][st][key], y[name][st][key], rtol=1e-12, atol=1e-15)
                    else:                       # per-run records: the same numbers
                        np.testing.assert_array_equal(x[name][st][key], y[name][st][key])
