"""MonteCarloJob.spread_outputs: moving an output region to another place in the device memory (found by timing: the launch is
faster when its planes lie in two of the GPU's three 96 GB thirds) must not change a result, leak a region or leave a stale
pointer behind -- whichever way the search ends."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import ginsim
    c = ginsim.Context(0)
    yield c
    c.close()


def _job(ctx, runs=4096, n=200, **kw):
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    truth = {k: (v[:n] if hasattr(v, 'shape') and v.shape and v.shape[0] >= n else v) for k, v in truth.items()}
    acc, gyr = workloads.imu_grade('mid-accuracy')
    return ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=runs, seed=21, **kw)


def _snapshot(job, algos):
    ids = np.array([0, 1, job.runs // 2, job.runs - 1])
    out = [job.sensors('accel', ids), job.sensors('gyro', ids)]
    for a in algos:
        out += list(job.trajectories(a, ids)) + [job.end_errors(a)]
    return out


@pytest.mark.parametrize('algos', [('free',), ('free', 'odo')])
def test_spread_outputs_keeps_every_result_whichever_way_the_search_ends(ctx, algos):
    kw = dict(algos=algos, keep_sensors=True, keep_traj=True)
    if 'odo' in algos:
        kw['odo_err'] = {'scale': 0.99, 'stdv': 0.1}
    job = _job(ctx, **kw)
    job.SPREAD_MIN = 1 << 20            # the regions of this small job count as large
    job.run()
    want = _snapshot(job, algos)
    ctx.release_pool()
    free0 = ctx.mem_info()[0]
    # no candidate can be 50 % faster: everything tried is freed, the original regions are bound again
    r = job.spread_outputs(tries=3, gain=0.5, min_gain=0.5, launches=2, good_bytes_per_s=None)
    assert r['moved'] is None and r['candidates'] == 6 and r['region'].startswith('traj_') and len(r['regions']) == 2, r      # two regions, three candidates each
    ctx.sync()
    for a, b in zip(want, _snapshot(job, algos)):
        np.testing.assert_array_equal(a, b)
    assert abs(ctx.mem_info()[0] - free0) <= (8 << 20), (free0, ctx.mem_info()[0])
    # every candidate "wins" (gain < 0): the first one stays, the original region is freed
    key = r['region']
    old = job.buffer(key).ptr
    r = job.spread_outputs(tries=3, gain=-10.0, min_gain=-10.0, launches=2, good_bytes_per_s=None)
    assert r['moved'] == key and r['candidates'] == 1 and r['launch_ms'] > 0, r
    assert job.buffer(key).ptr != old
    ctx.sync()
    for a, b in zip(want, _snapshot(job, algos)):
        np.testing.assert_array_equal(a, b)
    assert abs(ctx.mem_info()[0] - free0) <= (8 << 20), (free0, ctx.mem_info()[0])
    # and the job goes on as before: another batch, statistics of it
    job.params.run_offset = 10 ** 6
    job.run()
    assert job.stats('free').count == job.runs
    job.release()


def test_spread_outputs_moves_the_sensor_series_when_they_are_the_largest_region(ctx):
    """given sensors + trajectories: the second region is the input; sensors only + odometer: 'imu' moves, its views follow."""
    job = _job(ctx, runs=2048, algos=(), keep_sensors=True, odo_err={'scale': 1.0, 'stdv': 0.05})
    job.SPREAD_MIN = 1 << 18
    job.run()
    ids = np.arange(0, 2048, 97)
    want = [job.sensors(nm, ids) for nm in ('accel', 'gyro', 'odo')]
    r = job.spread_outputs(tries=2, gain=-10.0, min_gain=-10.0, launches=2, good_bytes_per_s=None)
    assert r['moved'] == 'imu', r
    assert job.buffer('accel').ptr == job.buffer('imu').ptr and job.buffer('gyro').ptr == job.buffer('imu').ptr + job.buffer('imu').nbytes // 2
    for a, nm in zip(want, ('accel', 'gyro', 'odo')):
        np.testing.assert_array_equal(a, job.sensors(nm, ids))
    job.release()


def test_spread_outputs_leaves_a_job_with_one_region_alone(ctx):
    job = _job(ctx, algos=('free',), keep_traj=True)
    job.SPREAD_MIN = 1 << 20
    job.run()
    assert job.spread_outputs()['moved'] is None
    stats_only = _job(ctx, algos=('free',))
    stats_only.run()
    assert stats_only.spread_outputs()['why'].startswith('fewer than two')
    job.release()
    stats_only.release()


def test_sim_with_spread_outputs_gives_the_same_results():
    import contextlib
    import io
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    from ginsim import workloads
    ini = np.array([32.0, 120.0, 0, 5, 0, 0, 0, 0, 0], dtype=float)
    ini[0:2] *= np.pi / 180
    out = []
    for spread in (False, True):
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
        sim = ins_sim.Sim([100.0, 0.0, 0.0], workloads.profile_path('turn_90deg'), ref_frame=1, imu=imu,
                          algorithm=free_integration.FreeIntegration(ini), seed=5, keep_trajectories=True, device=0, spread_outputs=spread)
        sim.run(64)
        with contextlib.redirect_stdout(io.StringIO()):
            sim.results(err_stats_start=-1)
        out.append(sim.err_stats)
    for k in ('att_euler', 'pos', 'vel'):
        for stat in ('max', 'avg', 'std'):
            a, b = out[0][k][stat], out[1][k][stat]
            if hasattr(a, 'keys'):
                assert sorted(a) == sorted(b)
                for kk in a:
                    np.testing.assert_array_equal(np.asarray(a[kk]), np.asarray(b[kk]))
            else:
                np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
