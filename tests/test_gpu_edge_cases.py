"""Edge cases of the hot path: degenerate sizes, ragged batches, multi-column initial states, 64-bit run ids,
argument validation through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
D2R = np.pi / 180


@pytest.fixture(scope='module')
def ctx():
    import ginsim
    c = ginsim.Context(0)
    yield c
    c.close()


@pytest.fixture(scope='module')
def turn():
    from ginsim import workloads
    out = {}
    for rf in (0, 1):
        ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
        out[rf] = (ini, truth)
    return out


def _cut(truth, n):
    return {k: (v[:n] if hasattr(v, 'shape') and v.shape and v.shape[0] >= n else v) for k, v in truth.items()}


@pytest.mark.parametrize('n', [1, 2, 3, 33])
@pytest.mark.parametrize('rf', [0, 1])
def test_tiny_sample_counts(ctx, turn, n, rf):
    """n = 1: only the initial state exists (free_integration.py:96-102); n = 2: one step."""
    import ginsim
    from ginsim import workloads
    from oracle import c_oracle
    ini, truth = turn[rf]
    t = _cut(truth, n)
    acc, gyr = workloads.imu_grade('low-accuracy')
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, t, acc, gyr, ini, runs=5, algos=('free', 'odo'),
                               odo_err={'scale': 0.99, 'stdv': 0.1}, seed=8, keep_sensors=True, keep_traj=True).run()
    for a in ('free', 'odo'):
        end, traj, sens = c_oracle.mc_run(8, 0, 5, 100.0, rf, t, acc, gyr, ini, algo=a, odo_err={'scale': 0.99, 'stdv': 0.1}, keep=5)
        att, pos, vel = job.trajectories(a, np.arange(5))
        assert att.shape == (5, n, 3)
        np.testing.assert_allclose(vel, traj[:, :, 6:9], rtol=0, atol=1e-11)
        np.testing.assert_allclose(pos, traj[:, :, 3:6], rtol=1e-14, atol=1e-9)
        np.testing.assert_allclose(job.end_errors(a)[:, 6:9], end[:, 6:9], rtol=0, atol=1e-11)
        np.testing.assert_allclose(job.sensors('accel', np.arange(5)), sens[:, :, 0:3], rtol=0, atol=1e-12)
    job.release()


def test_single_run_and_wave_boundaries(ctx, turn):
    import ginsim
    from ginsim import workloads
    ini, truth = turn[1]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    ref = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=257, seed=3).run().end_errors('free')
    for R in (1, 63, 64, 65, 255, 256):
        e = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=3).run().end_errors('free')
        np.testing.assert_array_equal(e, ref[:R])
    st = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=1, seed=3).run().stats('free')
    assert st.count == 1 and np.all(st.std == 0) and np.allclose(st.mean, ref[0])


def test_run_ids_beyond_32_bits(ctx, turn):
    import ginsim
    from ginsim import workloads
    from oracle import c_oracle
    ini, truth = turn[1]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    off = 2 ** 40 + 12345
    dev = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=70, seed=2 ** 63 + 9, run_offset=off).run().end_errors('free')
    end, _, _ = c_oracle.mc_run(2 ** 63 + 9, off, 70, 100.0, 1, truth, acc, gyr, ini)
    np.testing.assert_allclose(dev[:, 6:9], end[:, 6:9], rtol=0, atol=1e-10)
    low = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=70, seed=2 ** 63 + 9, run_offset=12345).run().end_errors('free')
    assert np.abs(dev - low).max() > 1e-6          # the high word of the run id really enters the counter


@pytest.mark.parametrize('rf', [0, 1])
def test_multi_column_initial_states_and_run_times(ctx, turn, rf):
    """free_integration.py:42-61, 85-87: run r uses column r while r < k, else column 0; run_times keeps counting."""
    import ginsim
    from ginsim import workloads
    from oracle import c_oracle
    ini, truth = turn[rf]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    k = 5
    table = np.repeat(ini[:, None], k, axis=1)
    table[3, :] += np.arange(k) * 0.5           # different initial forward speed per column
    table[6, :] += np.arange(k) * 1e-3          # and yaw
    g = np.full((1, k), 9.79)
    table10 = np.vstack([table, g])
    for tab in (table, table10):
        dev = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, tab, runs=9, seed=4, ini_first=2).run().end_errors('free')
        end, _, _ = c_oracle.mc_run(4, 0, 9, 100.0, rf, truth, acc, gyr, tab, ini_first=2)
        np.testing.assert_allclose(dev[:, 6:9], end[:, 6:9], rtol=0, atol=1e-9)
        assert np.abs(dev[0, 6:9] - dev[5, 6:9]).max() > 1e-3      # column 2 vs column 0


def test_sim_second_run_keeps_counting_columns(ctx, turn):
    import os
    from conftest import PKG
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    ini, _ = turn[1]
    table = np.repeat(ini[:, None], 4, axis=1)
    table[3, :] += np.arange(4)
    algo = free_integration.FreeIntegration(table)
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='high-accuracy', axis=6, gps=False)
    ends = []
    for _ in range(2):
        sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=algo, seed=1)
        sim.run(3)
        ends.append(np.stack([sim.dmgr.vel.data['algo0_%d' % r][-1] for r in range(3)]))
    assert algo.run_times == 6
    sp = lambda v: np.linalg.norm(v, axis=1)
    # an initial body-speed error d rotates with the vehicle through the 90-degree turn: |v| = sqrt(10^2 + d^2)
    np.testing.assert_allclose(sp(ends[0]), np.sqrt(100.0 + np.array([0.0, 1.0, 4.0])), atol=2e-3)     # columns 0,1,2
    np.testing.assert_allclose(sp(ends[1]), np.sqrt(100.0 + np.array([9.0, 0.0, 0.0])), atol=2e-3)     # column 3, then column 0


def test_white_drift_axes_mixed(ctx, turn):
    """b_corr = inf on some axes only (pathgen.py:591-593)."""
    import ginsim
    from oracle import ins_np
    ini, truth = turn[1]
    acc = {'b': np.array([1e-3, 0, 0]), 'b_drift': np.array([1e-3, 2e-3, 3e-3]), 'b_corr': np.array([np.inf, 50.0, np.inf]),
           'vrw': np.array([1e-3, 1e-3, 1e-3])}
    gyr = {'b': np.zeros(3), 'b_drift': np.array([1e-4, 1e-4, 1e-4]), 'b_corr': np.array([100.0, np.inf, 100.0]),
           'arw': np.array([1e-4, 1e-4, 1e-4])}
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=3, seed=6, keep_sensors=True).run()
    a, g = ins_np.mc_sensors(6, np.arange(3), 100.0, truth['ref_accel'], truth['ref_gyro'], acc, gyr)
    np.testing.assert_allclose(job.sensors('accel', [0, 1, 2]), a, rtol=0, atol=1e-12)
    np.testing.assert_allclose(job.sensors('gyro', [0, 1, 2]), g, rtol=0, atol=1e-14)


def test_argument_validation(ctx, turn):
    import ginsim
    from ginsim import workloads
    ini, truth = turn[1]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    with pytest.raises(ValueError):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4, algos=('odo',))          # no odometer model
    with pytest.raises(ValueError):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4, algos=('kalman',))
    bad = dict(acc)
    bad['vrw'] = np.array([np.nan, 0, 0])
    with pytest.raises(ValueError, match='non-finite'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, bad, gyr, ini, runs=4).run()
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4)
    job.params.ref_frame = 3
    with pytest.raises(ValueError, match='ref_frame'):
        job.run()
    with pytest.raises(ValueError):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4).run().trajectories('free', [0])   # not kept
    with pytest.raises(ValueError, match='out of range'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=4, keep_traj=True).run().trajectories('free', [4])
    # empty batches are refused by the C ABI, not launched (a rank without runs simply has no job: ginsim.distributed.shard)
    with pytest.raises(ValueError, match='must be >= 1'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=0).run()
    none = {k: (v[:0] if hasattr(v, 'shape') and v.ndim >= 1 and v.shape[0] > 1 else v) for k, v in truth.items()}
    with pytest.raises((ValueError, IndexError)):
        ginsim.MonteCarloJob(ctx, 100.0, 1, none, acc, gyr, ini, runs=4).run()


@pytest.mark.parametrize('rf,algos,keep,precision', [(1, ('free',), True, 'f64'), (0, ('free', 'odo'), False, 'f64'),
                                                     (1, ('free', 'odo'), True, 'f64'), (1, ('free',), True, 'f32'),
                                                     (0, ('free', 'odo'), True, 'f32')])
def test_plain_and_wave_specialised_kernels_agree_bitwise(ctx, turn, rf, algos, keep, precision):
    """Batches of <= 1024 wavefronts use the producer/consumer kernel, larger ones the one-wave-per-run-group kernel.
    Same runs through both must give identical bits (66 560 runs = 1040 wavefronts vs two halves of 520)."""
    import ginsim
    from ginsim import workloads
    ini, truth = turn[rf]
    t = {k: (v[:60] if hasattr(v, 'shape') and v.shape and v.shape[0] == 1000 else v) for k, v in truth.items()}
    acc, gyr = workloads.imu_grade('mid-accuracy')
    kw = dict(algos=algos, odo_err={'scale': 0.999, 'stdv': 0.1}, seed=12, keep_sensors=keep, keep_traj=keep,
              precision=precision)
    R = 66560
    big = ginsim.MonteCarloJob(ctx, 100.0, rf, t, acc, gyr, ini, runs=R, **kw).run()
    halves = [ginsim.MonteCarloJob(ctx, 100.0, rf, t, acc, gyr, ini, runs=R // 2, run_offset=h * (R // 2), ini_first=h * (R // 2), **kw).run()
              for h in range(2)]
    for a in algos:
        e = big.end_errors(a)
        np.testing.assert_array_equal(np.vstack([h.end_errors(a) for h in halves]), e)
        if keep:
            pick = [0, 33279, 33280, R - 1]
            for x, y in zip(big.trajectories(a, pick), [np.concatenate([halves[0].trajectories(a, [0, 33279])[k],
                                                                        halves[1].trajectories(a, [0, 33279])[k]]) for k in range(3)]):
                np.testing.assert_array_equal(x, y)
    if keep:
        np.testing.assert_array_equal(big.sensors('gyro', [5, 40000]),
                                      np.concatenate([halves[0].sensors('gyro', [5]), halves[1].sensors('gyro', [40000 - 33280])]))
    for j in [big] + halves:
        j.release()


@pytest.mark.parametrize('rf,algos', [(1, ('free',)), (0, ('free', 'odo'))])
def test_device_resident_given_sensors_replay_bitwise(ctx, turn, rf, algos):
    """Sensors materialised by one job, integrated again by a given-sensors job (the plugin boundary for a whole batch,
    ins_algo_manager.py:90): the mechanisation alone must reproduce the fused kernel's trajectories bit for bit."""
    import ginsim
    from ginsim import workloads
    ini, truth = turn[rf]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    R = 1500
    gen = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, algos=algos, odo_err={'scale': 0.999, 'stdv': 0.1},
                               seed=3, keep_sensors=True, keep_traj=True).run()
    given = {k: gen.buffer(k) for k in (('gyro', 'accel', 'odo') if 'odo' in algos else ('gyro', 'accel'))}
    rep = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, None, None, ini, runs=R, algos=algos, keep_traj=True, given=given).run()
    assert 'true' in rep.kernel_name()
    pick = [0, 1, 63, 64, 777, R - 1]
    for a in algos:
        np.testing.assert_array_equal(rep.end_errors(a), gen.end_errors(a))
        for x, y in zip(rep.trajectories(a, pick), gen.trajectories(a, pick)):
            np.testing.assert_array_equal(x, y)
        s1, s2 = rep.stats(a), gen.stats(a)
        np.testing.assert_array_equal(s1.pack(), s2.pack())
    with pytest.raises(ValueError, match='given sensors'):
        ginsim.MonteCarloJob(ctx, 100.0, rf, truth, None, None, ini, runs=R, algos=algos, given={'gyro': given['gyro']} if 'free' in algos else {})
    rep.release()
    gen.release()


def test_async_stats_slots_match_blocking_reduction(ctx, turn):
    """ginsim_end_stats_begin/_finish: the record of batch k is picked up after batch k+1 was enqueued (the overlap
    bench.py uses) and equals the blocking reduction of the same batch."""
    import ginsim
    from ginsim import workloads
    ini, truth = turn[1]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=3000, seed=9)
    blocking = []
    for k in range(3):
        job.params.run_offset = 3000 * k
        blocking.append(job.run().stats('free').pack())
    got = []
    for k in range(3):
        job.params.run_offset = 3000 * k
        job.launch()
        if k:
            got.append(job.stats_finish((k - 1) & 1).pack())
        job.stats_begin('free', k & 1)
    got.append(job.stats_finish(0).pack())
    np.testing.assert_array_equal(np.array(got), np.array(blocking))
    with pytest.raises(ValueError, match='nothing was begun'):
        job.stats_finish(3)
    job.stats_begin('free', 5)
    with pytest.raises(ValueError, match='still pending'):
        job.stats_begin('free', 5)
    job.stats_finish(5)
    with pytest.raises(ValueError):
        job.stats_begin('free', 8)
    job.release()


@pytest.mark.parametrize('runs,n,with_odo', [(1, 60000, False), (3, 20000, True), (700, 2500, False)])
def test_time_parallel_sensor_series_match_oracle_and_lane_per_run_kernel(ctx, runs, n, with_odo):
    """Sensors only, few runs, long series (Sim.run(1) as a data generator, the Allan flow): ginsim_mc_run cuts the time
    axis into chunks (Gauss-Markov carry by a linear scan).  Same normals -> same series as the oracle's sequential
    recurrence and as the lane-per-run kernel (taken here from a batch too large for the time-parallel path)."""
    import ginsim
    from ginsim import workloads
    from oracle import ins_np
    ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0)
    t = {k: (v[:n] if hasattr(v, 'shape') and v.shape and v.shape[0] > n else v) for k, v in truth.items()}
    acc, gyr = workloads.imu_grade('mid-accuracy')
    odo_err = {'scale': 0.998, 'stdv': 0.07} if with_odo else None
    job = ginsim.MonteCarloJob(ctx, 200.0, 0, t, acc, gyr, None, runs=runs, algos=(), odo_err=odo_err, seed=77,
                               keep_sensors=True).run()
    pick = sorted({0, runs // 2, runs - 1})
    a_dev, g_dev = job.sensors('accel', pick), job.sensors('gyro', pick)
    a_ref, g_ref = ins_np.mc_sensors(77, np.array(pick), 200.0, t['ref_accel'], t['ref_gyro'], acc, gyr)
    np.testing.assert_allclose(a_dev, a_ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g_dev, g_ref, rtol=0, atol=1e-14)
    if with_odo:
        o_ref = ins_np.mc_odo(77, np.array(pick), t['ref_odo'], odo_err)
        np.testing.assert_allclose(job.sensors('odo', pick).reshape(len(pick), -1), o_ref, rtol=0, atol=1e-12)
    big = ginsim.MonteCarloJob(ctx, 200.0, 0, t, acc, gyr, None, runs=1100, algos=(), odo_err=odo_err, seed=77,
                               keep_sensors=True).run()            # > 1024 runs: one lane per run
    if runs <= 1100:
        np.testing.assert_allclose(a_dev, big.sensors('accel', pick), rtol=0, atol=4e-15)      # an ulp of the terms
        np.testing.assert_allclose(g_dev, big.sensors('gyro', pick), rtol=0, atol=2e-16)
    big.release()
    job.release()


def test_pinned_host_buffers(ctx):
    """ginsim.pinned_empty (ginsim_host_alloc): page-locked NumPy arrays in and out of the host-buffer boundary give the
    same bits as pageable ones, and are released with the array."""
    import gc
    import ginsim
    g = load_golden('t1_fixture_bosch')
    gyro, accel = ginsim.pinned_empty(ctx, (8,) + g['gyro'].shape), ginsim.pinned_empty(ctx, (8,) + g['accel'].shape)
    assert gyro.flags['C_CONTIGUOUS'] and gyro.dtype == np.float64 and gyro.shape == (8, g['gyro'].shape[0], 3)
    gyro[...] = g['gyro']
    accel[...] = g['accel']
    a0 = ginsim.free_integration_host(ctx, 'free', 1, float(g['fs']), np.array(gyro), np.array(accel), ini=g['ini'][:9])
    a1 = ginsim.free_integration_host(ctx, 'free', 1, float(g['fs']), gyro, accel, ini=g['ini'][:9], pinned_out=True)
    for x, y in zip(a0, a1):
        np.testing.assert_array_equal(x, y)
    view = a1[0][3]
    del a1, gyro, accel
    gc.collect()
    assert np.isfinite(view).all()          # a view keeps the pinned block alive
    small = ginsim.pinned_empty(ctx, 0)
    assert small.size == 0


@pytest.mark.parametrize('runs,n', [(2, 2048), (5, 4099), (1024, 2111)])
def test_time_parallel_series_general_sensor_model_and_ragged_lengths(ctx, runs, n):
    """The series path with the GENERAL sensor model -- a constant bias on every axis, one accelerometer and one gyro axis with an
    infinite correlation time (white drift, pathgen.py:593), a short correlation time (a = 0.5: the scan weights a^k underflow
    towards 0 inside a wavefront) -- at lengths that end inside a 64-sample step and at the smallest / largest sizes the path
    takes, against the oracle's sequential recurrence."""
    import ginsim
    from ginsim import workloads
    from oracle import ins_np
    ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0)
    t = {k: (v[:n] if hasattr(v, 'shape') and v.shape and v.shape[0] > n else v) for k, v in truth.items()}
    acc = {'b': np.array([0.01, -0.02, 0.03]), 'b_drift': np.array([5e-5, 8e-5, 2e-5]), 'b_corr': np.array([100.0, np.inf, 0.01]),
           'vrw': np.array([5e-4, 4e-4, 6e-4])}
    gyr = {'b': np.array([1e-4, 0.0, -2e-4]), 'b_drift': np.array([2e-5, 1e-5, 3e-5]), 'b_corr': np.array([np.inf, 50.0, 200.0]),
           'arw': np.array([7e-5, 7e-5, 9e-5])}
    job = ginsim.MonteCarloJob(ctx, 200.0, 0, t, acc, gyr, None, runs=runs, algos=(), seed=4242, keep_sensors=True).run()
    assert job.sensor_layout == 'series' and job.kernel_name() == 'ginsim::series_kernel<1>'
    pick = sorted({0, runs // 2, runs - 1})
    a_ref, g_ref = ins_np.mc_sensors(4242, np.array(pick), 200.0, t['ref_accel'], t['ref_gyro'], acc, gyr)
    np.testing.assert_allclose(job.sensors('accel', pick), a_ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(job.sensors('gyro', pick), g_ref, rtol=0, atol=1e-14)
    job.release()


def test_pathgen_sensor_generators_under_their_reference_names(ctx):
    """pathgen.acc_gen / gyro_gen / odo_gen / gps_gen / mag_gen (pathgen.py:441-661) with the reference's signatures, served by the
    device: one realisation of the error model over given truth, equal to the oracle's for the same key; np.random.seed makes
    a call repeatable as it does for the reference; random / sinusoidal vibration on the device, a PSD refused."""
    from gnss_ins_sim.pathgen import pathgen
    from ginsim import workloads
    from oracle import ins_np
    ini, truth, raw = workloads.truth_from_profile('turn_90deg', 100.0, 0, fs_gps=10.0, gps=True)
    acc, gyr = workloads.imu_grade('low-accuracy')
    zero = np.zeros_like(truth['ref_accel'])
    quiet = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, np.inf), 'arw': np.zeros(3), 'vrw': np.zeros(3)}
    a = pathgen.acc_gen(100.0, truth['ref_accel'], acc, seed=91)
    w = pathgen.gyro_gen(100.0, truth['ref_gyro'], gyr, seed=92)
    a_ref, _ = ins_np.mc_sensors(91, np.array([0]), 100.0, truth['ref_accel'], zero, acc, quiet)
    _, w_ref = ins_np.mc_sensors(92, np.array([0]), 100.0, zero, truth['ref_gyro'], quiet, gyr)
    assert a.shape == truth['ref_accel'].shape and w.shape == truth['ref_gyro'].shape
    np.testing.assert_allclose(a, a_ref[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(w, w_ref[0], rtol=0, atol=1e-14)
    odo_err = {'scale': 0.999, 'stdv': 0.1}
    o = pathgen.odo_gen(truth['ref_odo'], odo_err, seed=93)
    np.testing.assert_allclose(o, ins_np.mc_odo(93, np.array([0]), truth['ref_odo'], odo_err)[0], rtol=0, atol=1e-12)
    np.random.seed(5)
    a1 = pathgen.acc_gen(100.0, truth['ref_accel'], acc)
    np.random.seed(5)
    a2 = pathgen.acc_gen(100.0, truth['ref_accel'], acc)
    a3 = pathgen.acc_gen(100.0, truth['ref_accel'], acc)
    assert np.array_equal(a1, a2) and not np.array_equal(a2, a3)
    gps_err = {'stdp': np.array([5.0, 5.0, 7.0]), 'stdv': np.array([0.05, 0.05, 0.05])}
    gp = pathgen.gps_gen(truth['ref_gps'], gps_err, 0, seed=94)
    assert gp.shape == truth['ref_gps'].shape
    d = gp - truth['ref_gps']
    assert 1.0 < d[:, 2].std() < 20.0 and 1e-7 < d[:, 0].std() < 1e-5 and 0.01 < d[:, 3:6].std() < 0.2      # m, rad, m/s
    mag_err = {'si': np.eye(3) + 0.01, 'hi': np.array([1.0, -2.0, 3.0]), 'std': np.array([0.1, 0.1, 0.1])}
    ref_mag = np.tile(np.array([30.0, 2.0, 40.0]), (500, 1))
    mg = pathgen.mag_gen(ref_mag, mag_err, seed=95)
    np.testing.assert_allclose(mg.mean(0), (ref_mag[0] + mag_err['hi']).dot(mag_err['si'].T), atol=0.05)
    # vib_def as the reference's callers pass it (pathgen.py:447-462): random / sinusoidal on the device, a PSD refused
    vr = {'type': 'random', 'x': 0.3, 'y': 0.1, 'z': 0.2}
    vs = {'type': 'sinusoidal', 'x': 0.02, 'y': 0.01, 'z': 0.03, 'freq': 3.0}
    av = pathgen.acc_gen(100.0, truth['ref_accel'], acc, vr, seed=91)
    a_ref, _ = ins_np.mc_sensors(91, np.array([0]), 100.0, truth['ref_accel'], zero, acc, quiet, vib_accel=vr)
    np.testing.assert_allclose(av, a_ref[0], rtol=0, atol=1e-12)
    assert 0.25 < (av - a)[:, 0].std() < 0.35                      # same key: the difference IS the vibration
    wv = pathgen.gyro_gen(100.0, truth['ref_gyro'], gyr, vs, seed=92)
    _, w_ref = ins_np.mc_sensors(92, np.array([0]), 100.0, zero, truth['ref_gyro'], quiet, gyr, vib_gyro=vs)
    np.testing.assert_allclose(wv, w_ref[0], rtol=0, atol=1e-14)
    assert abs(np.abs(wv - w)[:, 2].max() - 0.03) < 1e-3            # a sine of the peak value given
    # a PSD (ABI 8; pathgen.py:479-484 -> time_series_from_psd.py): interpolated, and GIVEN on the series' grid -- the reference halves
    # that one in place at every call, so the second call of the same arrays sees a quarter
    vp = {'type': 'psd', 'freq': np.array([0.0, 10.0, 30.0]), 'x': np.array([1e-3, 4e-3, 1e-3]), 'y': np.full(3, 2e-3), 'z': np.full(3, 1e-3)}
    ap = pathgen.acc_gen(100.0, truth['ref_accel'], acc, vp, seed=91)
    a_ref, _ = ins_np.mc_sensors(91, np.array([0]), 100.0, truth['ref_accel'], zero, acc, quiet, vib_accel=vp)
    np.testing.assert_allclose(ap, a_ref[0], rtol=0, atol=1e-12)
    assert 0.2 < (ap - a).std() < 0.5 and vp['x'].tolist() == [1e-3, 4e-3, 1e-3]
    n = truth['ref_gyro'].shape[0]
    f = np.linspace(0.0, 50.0, n // 2 + 1)
    vg = {'type': 'psd', 'freq': f, 'x': np.full(f.shape, 1e-6), 'y': np.full(f.shape, 2e-6), 'z': 1e-6 * (1.0 + f / 50.0)}
    keep = {k: vg[k].copy() for k in 'xyz'}
    w1 = pathgen.gyro_gen(100.0, truth['ref_gyro'], gyr, vg, seed=92)
    _, w_ref = ins_np.mc_sensors(92, np.array([0]), 100.0, zero, truth['ref_gyro'], quiet, gyr, vib_gyro=dict(vg, **keep))
    np.testing.assert_allclose(w1, w_ref[0], rtol=0, atol=1e-14)
    assert np.array_equal(vg['y'][1:-1], 0.5 * keep['y'][1:-1]) and vg['y'][0] == keep['y'][0] and vg['y'][-1] == keep['y'][-1]
    w2 = pathgen.gyro_gen(100.0, truth['ref_gyro'], gyr, vg, seed=92)                       # the same key, the halved arrays
    assert 0.65 < (w2 - w).std() / (w1 - w).std() < 0.76                                    # sqrt(1/2) of the amplitude


VIB_CASES = {
    'random': ({'type': 'random', 'x': 0.2, 'y': 0.05, 'z': 0.1}, {'type': 'random', 'x': 2e-3, 'y': 1e-3, 'z': 3e-3}),
    'sinusoidal': ({'type': 'sinusoidal', 'x': 0.3, 'y': 0.1, 'z': 0.2, 'freq': 7.0},
                   {'type': 'sinusoidal', 'x': 5e-3, 'y': 2e-3, 'z': 1e-3, 'freq': 0.9}),
    'accel only': ({'type': 'sinusoidal', 'x': 0.0, 'y': 0.4, 'z': 0.0, 'freq': 21.5}, None),
    'gyro only': (None, {'type': 'random', 'x': 1e-3, 'y': 1e-3, 'z': 1e-3}),
}


@pytest.mark.parametrize('case', sorted(VIB_CASES))
@pytest.mark.parametrize('rf,algos', [(1, ('free',)), (0, ('free', 'odo')), (0, ('odo',))])
def test_vibration_models_against_the_oracles(ctx, turn, case, rf, algos):
    """Sim(env=...)'s vibration term in the fused kernel (ABI 5; pathgen.py:476-492, 538-556): a ragged batch with global run
    ids offset, sensors per sample and end-point errors against the NumPy and the C restatements; the launch is invariant to
    how the runs are split (counter RNG: the vibration normals and phases belong to the GLOBAL run id)."""
    import ginsim
    from ginsim import workloads
    from oracle import ins_np, c_oracle
    ini, truth = turn[rf]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    va, vg = VIB_CASES[case]
    odo_err = {'scale': 0.999, 'stdv': 0.1}
    R, off, seed = 150, 4000, 1234
    kw = dict(algos=algos, odo_err=odo_err, seed=seed, vib_accel=va, vib_gyro=vg)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, run_offset=off, keep_sensors=True, keep_traj=True, **kw).run()
    assert job.kernel_name().endswith('true>') and ginsim.lib.ginsim_mc_variant is not None
    runs = np.arange(off, off + R)
    a_ref, g_ref = ins_np.mc_sensors(seed, runs, 100.0, truth['ref_accel'], truth['ref_gyro'], acc, gyr, va, vg)
    pick = np.array([0, 63, 64, 149])
    np.testing.assert_allclose(job.sensors('accel', pick), a_ref[pick], rtol=0, atol=1e-12)
    np.testing.assert_allclose(job.sensors('gyro', pick), g_ref[pick], rtol=0, atol=1e-14)
    for a in algos:
        end, _, _ = c_oracle.mc_run(seed, off, R, 100.0, rf, truth, acc, gyr, ini, algo=a, odo_err=odo_err, vib_accel=va, vib_gyro=vg)
        got = job.end_errors(a)
        np.testing.assert_allclose(got[:, 3:], end[:, 3:], rtol=0, atol=2e-8)      # |pos| ~ 5e6 m in ref_frame 1
        assert np.max(np.abs(np.mod(got[:, :3] - end[:, :3] + np.pi, 2 * np.pi) - np.pi)) < 1e-9
    # the same runs as two launches (statistics only) give the same end-point errors to the bit
    parts = [ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=r, run_offset=off + o, **kw).run() for o, r in ((0, 70), (70, 80))]
    for a in algos:
        assert np.array_equal(np.concatenate([q.end_errors(a) for q in parts]), job.end_errors(a))
    for q in parts + [job]:
        q.release()


def _psd_def(unit=1.0, grid=None):
    """a vib_def of type 'psd' as Sim.__parse_env makes it from an (n, 4) array: six rows to interpolate, or `grid` rows on the grid
    of a series of 2 (grid - 1) samples at 100 Hz"""
    if grid is None:
        f = np.array([0.0, 1.0, 4.0, 11.0, 30.0, 50.0])
        x = unit * np.array([0.0, 2e-4, 8e-4, 3e-4, 1e-4, 2e-5])
    else:
        f = np.linspace(0.0, 50.0, grid)
        x = unit * (3e-4 * np.exp(-((f - 9.0) / 5.0) ** 2) + 2e-5)
    return {'type': 'psd', 'freq': f, 'x': x.copy(), 'y': 0.5 * x[::-1], 'z': x + unit * 1e-4}


@pytest.mark.parametrize('rf,algos', [(1, ('free',)), (0, ('free', 'odo'))])
def test_psd_vibration_against_the_oracle(ctx, turn, rf, algos):
    """Sim(env=<PSD array>) (ABI 8; pathgen.py:479-484, :541-546 -> time_series_from_psd.py): a ragged batch with global run ids
    offset -- the accelerometer's PSD interpolated, the gyroscope's given on the series' own grid (the reference halves that one
    in place at every run: run g sees 0.5^(g + 1)) -- sensors per sample and trajectories against the NumPy restatement, which
    the goldens hold to the reference; the launch is invariant to how the runs are split (the phases belong to the GLOBAL run id)."""
    import ginsim
    from ginsim import workloads
    from oracle import ins_np
    ini, truth = turn[rf]
    n = truth['ref_accel'].shape[0]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    va, vg = _psd_def(), _psd_def(1e-3, grid=n // 2 + 1)
    before = vg['x'].copy()
    odo_err = {'scale': 0.999, 'stdv': 0.1}
    R, off, seed = 150, 37, 99
    kw = dict(algos=algos, odo_err=odo_err, seed=seed, vib_accel=va, vib_gyro=vg)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=R, run_offset=off, keep_sensors=True, keep_traj=True, **kw).run()
    assert job.kernel_name().startswith('ginsim::mc_kernel<') and job.kernel_name().endswith('true>') and job.psd_given_on_grid
    assert np.array_equal(vg['x'], before)                  # the engine mutates nothing (the drop-in Sim does, as the reference)
    runs = np.arange(off, off + R)
    a_ref, g_ref = ins_np.mc_sensors(seed, runs, 100.0, truth['ref_accel'], truth['ref_gyro'], acc, gyr, va, vg)
    pick = np.array([0, 63, 64, 149])
    np.testing.assert_allclose(job.sensors('accel', pick), a_ref[pick], rtol=0, atol=1e-12)
    np.testing.assert_allclose(job.sensors('gyro', pick), g_ref[pick], rtol=0, atol=1e-14)
    a_plain, _ = ins_np.mc_sensors(seed, runs[pick], 100.0, truth['ref_accel'], truth['ref_gyro'], acc, gyr)
    assert np.abs(a_ref[pick] - a_plain).std() > 0.05       # the term is there: ~0.1 m/s^2 rms
    odo = ins_np.mc_odo(seed, runs[pick], truth['ref_odo'], odo_err) if 'odo' in algos else None
    for a in algos:
        att, pos, vel = ins_np.free_integration(rf, 100.0, g_ref[pick], a_ref[pick], ini, odo=odo if a == 'odo' else None)
        got = job.trajectories(a, pick)
        np.testing.assert_allclose(got[2], vel, rtol=0, atol=1e-9)
        np.testing.assert_allclose(got[1], pos, rtol=1e-12, atol=1e-8)
        assert np.max(np.abs(np.mod(got[0] - att + np.pi, 2 * np.pi) - np.pi)) < 1e-10
    # the same runs as two launches: their series come out of FFT batches of other sizes -- equal to rounding, not to the bit
    parts = [ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=r, run_offset=off + o, **kw).run() for o, r in ((0, 70), (70, 80))]
    for a in algos:
        np.testing.assert_allclose(np.concatenate([q.end_errors(a) for q in parts])[:, 3:], job.end_errors(a)[:, 3:], rtol=1e-9, atol=1e-9)
    for q in parts + [job]:
        q.release()


def test_psd_vibration_series_in_blocks_tiled_and_refused(ctx, turn):
    """ginsim_vib_psd_series makes its series in blocks of runs (two buffers of about 256 MiB): 700 runs of a 16384-point period
    are two blocks (640 + 60); a series longer than 16384 samples repeats the period (time_series_from_psd.py:41-43, :58-63); a PSD
    that reaches beyond fs / 2 gives the reference's zeros (:32-34) -- no term at all, the bits of the launch without an
    environment; what the kernels do not carry is refused."""
    import ctypes as C
    import ginsim
    from ginsim import workloads, _lib
    from oracle import ins_np
    ini, truth = turn[1]
    acc, gyr = workloads.imu_grade('low-accuracy')
    n = 16384 + 600
    long_truth = {k: (np.concatenate([v] * 17)[:n] if hasattr(v, 'shape') and v.shape and v.shape[0] == 1000 else v) for k, v in truth.items()}
    va = _psd_def()
    R, seed = 700, 5
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, long_truth, acc, gyr, None, runs=R, algos=(), seed=seed, keep_sensors=True, vib_accel=va).run()
    assert job.params.vib_accel.period == 16384 and job.params.vib_accel.type == 3 and job.sensor_layout == 'runs'
    pick = np.array([0, 639, 640, 699])
    a_ref, g_ref = ins_np.mc_sensors(seed, pick, 100.0, long_truth['ref_accel'], long_truth['ref_gyro'], acc, gyr, va, None)
    got = job.sensors('accel', pick)
    np.testing.assert_allclose(got, a_ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(job.sensors('gyro', pick), g_ref, rtol=0, atol=1e-14)
    vib = got - ins_np.mc_sensors(seed, pick, 100.0, long_truth['ref_accel'], long_truth['ref_gyro'], acc, gyr)[0]
    np.testing.assert_allclose(vib[:, 16384:], vib[:, :600], rtol=0, atol=1e-12)          # the period repeats
    job.release()
    # beyond fs / 2: zeros
    ini1, t1 = turn[1]
    far = _psd_def()
    far['freq'] = far['freq'] * 1.2
    a = ginsim.MonteCarloJob(ctx, 100.0, 1, t1, acc, gyr, ini1, runs=64, seed=3, keep_sensors=True, vib_gyro=far).run()
    b = ginsim.MonteCarloJob(ctx, 100.0, 1, t1, acc, gyr, ini1, runs=64, seed=3, keep_sensors=True).run()
    assert a.params.vib_gyro.type == 0 and np.array_equal(a.sensors('gyro', np.arange(64)), b.sensors('gyro', np.arange(64)))
    # refusals
    with pytest.raises(NotImplementedError, match='fp64'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, t1, acc, gyr, ini1, runs=64, precision='f32', vib_accel=va)
    with pytest.raises(ValueError, match='given sensors'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, t1, None, None, ini1, runs=64, vib_accel=va, given={'gyro': a.buffer('gyro'), 'accel': a.buffer('accel')})
    b.params.vib_accel.type = 3                                 # type 3 without its series
    with pytest.raises(ValueError, match='series'):
        b.run()
    buf = ctx.malloc(3 * 8 * 64 * 8)
    amp = np.ones((3, 5))
    for period, runs_, sensor in ((7, 64, 0), (16386, 64, 0), (8, 0, 0), (8, 64, 2)):
        with pytest.raises(ValueError):
            _lib.check(ginsim.lib.ginsim_vib_psd_series(ctx.handle, amp.ctypes.data, period, runs_, 0, 1, sensor, 0, buf.ptr))
    amp[1, 2] = -1.0
    with pytest.raises(ValueError, match='amplitude'):
        _lib.check(ginsim.lib.ginsim_vib_psd_series(ctx.handle, amp.ctypes.data, 8, 64, 0, 1, 0, 0, buf.ptr))
    # the smallest period: two samples, bins 0 and 1 (both real in the inverse transform)
    amp = np.array([[4.0, 2.0], [0.0, 6.0], [8.0, 0.0]])
    _lib.check(ginsim.lib.ginsim_vib_psd_series(ctx.handle, amp.ctypes.data, 2, 64, 11, 7, 1, 0, buf.ptr))
    x = ctx.download(buf, (3, 2, 64))
    from oracle import philox
    for r in (0, 63):
        z = philox.vib_normals(7, 11 + r, 2, 'gyr')
        c = np.cos(np.pi * z)                                   # (bin, axis)
        want = np.stack([(amp[:, 0] * c[0] + amp[:, 1] * c[1]) / 2.0, (amp[:, 0] * c[0] - amp[:, 1] * c[1]) / 2.0], axis=1)
        np.testing.assert_allclose(x[:, :, r], want, rtol=0, atol=1e-14)
    buf.free()
    for q in (a, b):
        q.release()


@pytest.mark.parametrize('case', sorted(VIB_CASES))
@pytest.mark.parametrize('runs,n', [(1, 50000), (6, 4099)])
def test_vibration_on_the_time_parallel_series_kernels(ctx, case, runs, n):
    """Few runs, long series, Sim(env=...) -- the data generator of the Allan flow with a vibration environment: a per-sample term,
    so pass B of the series kernels carries it (series_kernel<2>); against the oracle and against the lane-per-run kernel."""
    import ginsim
    from ginsim import workloads
    from oracle import ins_np
    ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0)
    t = {k: (v[:n] if hasattr(v, 'shape') and v.shape and v.shape[0] > n else v) for k, v in truth.items()}
    acc, gyr = workloads.imu_grade('low-accuracy')
    va, vg = VIB_CASES[case]
    job = ginsim.MonteCarloJob(ctx, 200.0, 0, t, acc, gyr, None, runs=runs, run_offset=9, algos=(), seed=31, keep_sensors=True,
                               vib_accel=va, vib_gyro=vg).run()
    assert job.sensor_layout == 'series' and job.kernel_name() == 'ginsim::series_kernel<2>'
    ids = np.arange(runs)
    a_ref, g_ref = ins_np.mc_sensors(31, 9 + ids, 200.0, t['ref_accel'], t['ref_gyro'], acc, gyr, va, vg)
    np.testing.assert_allclose(job.sensors('accel', ids), a_ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(job.sensors('gyro', ids), g_ref, rtol=0, atol=1e-14)
    big = ginsim.MonteCarloJob(ctx, 200.0, 0, t, acc, gyr, None, runs=1100, run_offset=9, algos=(), seed=31, keep_sensors=True,
                               vib_accel=va, vib_gyro=vg).run()     # > 1024 runs: one lane per run
    assert big.kernel_name() == 'ginsim::mc_kernel<0, 0, false, true, 0, true>'
    np.testing.assert_allclose(job.sensors('accel', ids), big.sensors('accel', ids), rtol=0, atol=4e-15)
    np.testing.assert_allclose(job.sensors('gyro', ids), big.sensors('gyro', ids), rtol=0, atol=2e-16)
    if runs == 1:       # and through the Allan call of the job (series-major: no re-layout)
        tau, ad = job.allan(names=('gyro',))
        assert ad['gyro'].shape == (1, tau.size, 3) and np.all(np.isfinite(ad['gyro']))
    big.release()
    job.release()


def test_vibration_is_refused_where_it_does_not_live(ctx, turn):
    import ginsim
    from ginsim import workloads
    ini, truth = turn[1]
    acc, gyr = workloads.imu_grade('mid-accuracy')
    v = {'type': 'random', 'x': 0.1, 'y': 0.1, 'z': 0.1}
    given = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=64, keep_sensors=True).run()
    with pytest.raises(ValueError, match='given sensors'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, None, None, ini, runs=64, vib_accel=v,
                             given={'gyro': given.buffer('gyro'), 'accel': given.buffer('accel')})
    given.release()
    with pytest.raises(ValueError, match='unknown vibration type'):
        ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=64, vib_gyro={'type': 'square', 'x': 1, 'y': 1, 'z': 1})
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=64, vib_accel=v)
    job.params.vib_accel.type = 7
    with pytest.raises(ValueError, match='vibration type'):
        job.launch()
    job.params.vib_accel.type = 1
    job.params.vib_accel.amp[1] = float('nan')
    with pytest.raises(ValueError, match='finite'):
        job.launch()
    job.release()


@pytest.mark.parametrize('rf', [0, 1])
def test_vibration_in_the_wave_specialised_kernel_is_bit_identical(rf, monkeypatch):
    """Round 5: batches of at most 1024 wavefronts of runs (C2's shape) with a vibration environment run on
    mc_kernel_split<RF, 1, true, 1, true, VIB = true> -- the 'random' normals from the producers through the LDS ring, a sinusoidal
    term in the consumer -- instead of the plain vibration kernel with one wavefront per SIMD.  Same operations on the same values
    (pathgen.py:476-492, 538-556 restated once, vibration_term): sensors, trajectories and end-point errors equal to the bit."""
    import ginsim
    from ginsim import workloads
    fs = 100.0
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('low-accuracy')
    ctx = ginsim.default_context()
    envs = [({'type': 'random', 'x': 0.3, 'y': 0.1, 'z': 0.2}, {'type': 'random', 'x': 0.01, 'y': 0.02, 'z': 0.03}),
            ({'type': 'sinusoidal', 'x': 0.3, 'y': 0.1, 'z': 0.2, 'freq': 2.5}, {'type': 'sinusoidal', 'x': 0.01, 'y': 0.02, 'z': 0.03, 'freq': 0.7}),
            ({'type': 'sinusoidal', 'x': 0.5, 'y': 0.1, 'z': 0.2, 'freq': 12.0}, {'type': 'random', 'x': 0.002, 'y': 0.001, 'z': 0.003}),
            (None, {'type': 'random', 'x': 0.002, 'y': 0.001, 'z': 0.003})]
    ids = np.array([0, 63, 64, 700, 1336])
    for va, vg in envs:
        got = {}
        for flag in ('1', '0'):
            monkeypatch.setenv('GINSIM_SPLIT_VIB', flag)
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=1337, seed=99, run_offset=11, keep_sensors=True, keep_traj=True,
                                       vib_accel=va, vib_gyro=vg)
            name = job.kernel_name()
            job.run()
            got[flag] = (name, job.sensors('accel', ids), job.sensors('gyro', ids)) + job.trajectories('free', ids) + (job.end_errors('free'),)
            job.release()
        assert got['1'][0] == 'ginsim::mc_kernel_split<%d, 1, true, 1, true, true>' % rf and got['0'][0] == 'ginsim::mc_kernel<%d, 1, false, true, 0, true>' % rf
        for a, b in zip(got['1'][1:], got['0'][1:]):
            np.testing.assert_array_equal(a, b)
