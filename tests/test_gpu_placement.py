"""Placed device memory (ABI 7, csrc/placed.hip): the materialised series of a job are carved from an arena whose stripes cycle
through the classes of physical memory of the GPU.  Where a region lies must not change a single bit of a result; the arena's
free list, growth and release must neither lose a region's contents nor leak memory; and the unconfigured Sim -- the call the
reference's users make, ins_sim.py:164-192 -- gets the placement by default."""
import contextlib
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 1 << 30


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


def test_the_arena_spans_the_classes_of_physical_memory(ctx):
    """An MI355X has three; two are the least an arena is built from (fewer: GINSIM_ERR_PLACED and plain hipMalloc)."""
    assert ctx.placed_reserve(3 * G), ctx.placed_note
    info = ctx.placed_info()
    assert info['available'] == 1 and info['classes'] >= 2, info
    assert info['mapped_bytes'] >= 3 * G and info['mapped_bytes'] % info['stripe_bytes'] == 0
    per = info['stripes_of_class']
    assert sum(per) == info['mapped_bytes'] // info['stripe_bytes'] and sum(1 for k in per if k) >= 2, info
    assert 4 * max(per) <= 3 * sum(per) + 3, info                   # no class holds more than three quarters
    assert set(info['stripe_classes']) <= set('ABC') and len(info['stripe_classes']) == sum(per)
    assert info['searches'] >= 1 and info['chunks_created'] >= sum(per) and 0 < info['search_seconds'] < 60 and info['anchor_ms'] > 0


@pytest.mark.parametrize('algos', [('free',), ('free', 'odo')])
def test_where_the_planes_lie_changes_no_result(ctx, algos):
    kw = dict(algos=algos, keep_sensors=True, keep_traj=True)
    if 'odo' in algos:
        kw['odo_err'] = {'scale': 0.99, 'stdv': 0.1}
    plain = _job(ctx, placed=False, **kw).run()
    placed = _job(ctx, placed=True, **kw).run()
    where = placed.placement()
    assert where['placed'] == sorted(['imu'] + (['odo'] if 'odo' in algos else []) + ['traj_' + a for a in algos]) and not where['unplaced'], where
    assert plain.placement()['placed'] == [] and not plain.buffer('imu').placed
    # views handed out before and after the launch point into the carved region
    imu = placed.buffer('imu')
    assert imu.placed and placed.buffer('accel').ptr == imu.ptr and placed.buffer('gyro').ptr == imu.ptr + imu.nbytes // 2
    for a, b in zip(_snapshot(plain, algos), _snapshot(placed, algos)):
        np.testing.assert_array_equal(a, b)
    for a in algos:
        sa, sb = plain.stats(a), placed.stats(a)
        assert sa.count == sb.count and np.array_equal(sa.m2, sb.m2) and np.array_equal(sa.maxabs, sb.maxabs)
    # the mechanisation alone on the placed sensors (the plugin's run(set_of_input) boundary): same bits again
    import ginsim
    from ginsim import workloads
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
    truth = {k: (v[:200] if hasattr(v, 'shape') and v.shape and v.shape[0] >= 200 else v) for k, v in truth.items()}
    rep = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, None, None, ini, runs=placed.runs, keep_traj=True, placed=True,
                               given={'gyro': placed.buffer('gyro'), 'accel': placed.buffer('accel')}).run()
    np.testing.assert_array_equal(rep.end_errors('free'), plain.end_errors('free'))
    for j in (rep, placed, plain):
        j.release()
    assert ctx.placed_info()['used_bytes'] == 0


def test_small_jobs_stay_with_hipmalloc_by_default(ctx):
    job = _job(ctx, algos=('free',), keep_sensors=True, keep_traj=True).run()
    assert job.placement()['placed'] == [] and sorted(job.placement()['unplaced']) == ['imu', 'traj_free']
    job.release()


def test_free_list_growth_and_release(ctx):
    import ginsim
    ctx.sync()
    ctx.release_pool()
    assert ctx.placed_info()['used_bytes'] == 0
    free0 = ctx.mem_info()[0]
    mib = 1 << 20
    a = ctx.malloc(300 * mib, placed=True)
    b = ctx.malloc(700 * mib, placed=True)
    c = ctx.malloc(100 * mib, placed=True)
    assert a.placed and b.placed and c.placed and len({a.ptr, b.ptr, c.ptr}) == 3
    pattern = np.arange(1 << 16, dtype=np.float64)
    ginsim.lib.ginsim_memcpy_h2d(ctx.handle, a.ptr, pattern.ctypes.data, pattern.nbytes)
    ginsim.lib.ginsim_memcpy_h2d(ctx.handle, c.at(100 * mib - pattern.nbytes), pattern.ctypes.data, pattern.nbytes)
    hole = b.ptr
    b.free()
    b2 = ctx.malloc(700 * mib, placed=True)
    assert b2.ptr == hole                                   # first fit: the hole that was just freed
    before = ctx.placed_info()
    big = ctx.malloc(before['mapped_bytes'] + 2 * before['stripe_bytes'], placed=True)      # cannot fit: the arena grows
    after = ctx.placed_info()
    assert big.placed and after['mapped_bytes'] > before['mapped_bytes'] and after['searches'] == before['searches'] + 1, (ctx.placed_note, before, after)
    # what was carved before the growth is where it was, contents and all
    np.testing.assert_array_equal(ctx.download(a, pattern.shape), pattern)
    np.testing.assert_array_equal(ctx.download(c.at(100 * mib - pattern.nbytes), pattern.shape), pattern)
    assert ginsim.lib.ginsim_memset(ctx.handle, big.ptr, 0, big.nbytes) == 0            # every byte of the grown range is writable
    ctx.sync()
    used = after['used_bytes']
    assert used >= a.nbytes + b2.nbytes + c.nbytes + big.nbytes
    with pytest.raises(ValueError):                         # not the start of a carved region
        ginsim._lib.check(ginsim.lib.ginsim_free(ctx.handle, a.ptr + (2 << 20)))
    for x in (a, b2, c, big):
        x.free()
    assert ctx.placed_info()['used_bytes'] == 0
    ctx.release_pool()                                       # nothing carved: the arena's memory goes back to the driver
    info = ctx.placed_info()
    assert info['mapped_bytes'] == 0 and info['available'] == 0
    import time
    for _ in range(50):                                      # the driver returns released chunks a moment later
        if ctx.mem_info()[0] >= free0 - (64 << 20):
            break
        time.sleep(0.1)
    assert ctx.mem_info()[0] >= free0 - (64 << 20), (free0, ctx.mem_info()[0])


def test_a_replaced_job_in_a_reference_cycle_gives_its_stripes_to_the_next(ctx):
    """A script that makes one Sim after another leaves the earlier ones unreachable but in reference cycles: their regions would
    stay carved until the collector happens to run, and every new job would grow the arena -- a search of 0.3 - 4 s each.
    Context.placed_reserve lets the collector run before it grows: the next job of the same size re-uses the stripes."""
    import gc
    ctx.sync()
    gc.collect()
    ctx.release_pool()
    first = _job(ctx, runs=16384, n=200, algos=('free',), keep_sensors=True, keep_traj=True, placed=True).run()
    info = ctx.placed_info()
    searches, mapped, used = info['searches'], info['mapped_bytes'], info['used_bytes']
    assert used > 0 and first.placement()['placed'] and mapped < 12 * used
    gc.disable()
    try:
        holder = {'job': first}
        holder['self'] = holder             # a cycle: dropping the names frees nothing by itself
        del first, holder
        assert ctx.placed_info()['used_bytes'] == used
        for _ in range(12):                 # twelve more jobs of that size do not fit what is mapped -- unless the dead ones let go
            nxt = _job(ctx, runs=16384, n=200, algos=('free',), keep_sensors=True, keep_traj=True, placed=True).run()
            assert nxt.placement()['placed']
            holder = {'job': nxt}
            holder['self'] = holder
            del nxt, holder
            info = ctx.placed_info()
            assert info['searches'] == searches and info['mapped_bytes'] == mapped, info
    finally:
        gc.enable()
    gc.collect()
    assert ctx.placed_info()['used_bytes'] == 0


def test_arena_is_rebuilt_again_and_again(ctx):
    """build -> carve -> write -> free -> give back, six times: the create / map / unmap / release sequences of the driver's
    virtual-memory API, with a launch writing every carved byte in between."""
    import ginsim
    for cycle in range(6):
        assert ctx.placed_reserve((2 + cycle % 3) * G), ctx.placed_note
        buf = ctx.malloc((2 + cycle % 3) * G - (64 << 20), placed=True)
        assert buf.placed
        assert ginsim.lib.ginsim_memset(ctx.handle, buf.ptr, cycle, buf.nbytes) == 0
        ctx.sync()
        tail = ctx.download(buf.at(buf.nbytes - 4096), (4096,), dtype=np.uint8)
        assert (tail == cycle).all()
        buf.free()
        ctx.release_pool()
        assert ctx.placed_info()['mapped_bytes'] == 0


def test_two_contexts_of_a_device_share_its_arena(ctx):
    import ginsim
    other = ginsim.Context(0)
    a = ctx.malloc(600 << 20, placed=True)
    b = other.malloc(600 << 20, placed=True)
    assert a.placed and b.placed and a.ptr != b.ptr
    ia, ib = ctx.placed_info(), other.placed_info()
    assert ia['mapped_bytes'] == ib['mapped_bytes'] and ia['used_bytes'] == ib['used_bytes'] >= a.nbytes + b.nbytes
    b.free()
    other.close()                                           # not the device's last context: the arena stays
    assert ctx.placed_info()['used_bytes'] >= a.nbytes and ctx.placed_info()['mapped_bytes'] == ia['mapped_bytes']
    a.free()


def _sim(R, **kw):
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    from ginsim import workloads
    ini = np.array([32.0, 120.0, 0, 5, 0, 0, 0, 0, 0], dtype=float)
    ini[0:2] *= np.pi / 180
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    sim = ins_sim.Sim([100.0, 0.0, 0.0], workloads.profile_path('turn_90deg'), ref_frame=1, imu=imu,
                      algorithm=free_integration.FreeIntegration(ini), seed=5, keep_trajectories=True, device=0, **kw)
    sim.run(R)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(err_stats_start=-1)
    return sim


def test_the_unconfigured_sim_gets_placed_planes_and_the_same_statistics():
    """Sim(...).run(16 384) materialises 1.97 GB: placed by default (no keyword, no environment variable); placed=False and the
    round-5 alias spread_outputs=False give hipMalloc planes; the statistics do not differ in a bit."""
    assert os.environ.get('GINSIM_PLACED', '1') != '0'
    sims = [_sim(16384), _sim(16384, placed=False), _sim(16384, spread_outputs=False)]
    assert sims[0].placement['placed'] == ['imu', 'traj_free'] and sims[0].placement['arena']['available'] == 1, sims[0].placement
    assert sims[0].placement['arena']['classes'] >= 2
    for s in sims[1:]:
        assert s.placement['placed'] == [] and s.placement['unplaced'] == ['imu', 'traj_free']
    for k in ('att_euler', 'pos', 'vel'):
        for stat in ('max', 'avg', 'std'):
            for s in sims[1:]:
                np.testing.assert_array_equal(np.asarray(sims[0].err_stats[k][stat]), np.asarray(s.err_stats[k][stat]))
    run7 = [s.dmgr.pos.data['algo0_7'] for s in sims[:2]]
    np.testing.assert_array_equal(run7[0], run7[1])
    small = _sim(64)
    assert small.placement['placed'] == []


def test_no_usable_arena_means_plain_hipmalloc_not_an_error(ctx):
    """A search whose budget cannot hold the request (here: three chunks), or that cannot tell two classes apart, ends as GINSIM_ERR_PLACED; the
    context remembers why, jobs fall back to plain hipMalloc planes and compute the same bits.  A limit smaller than a request
    likewise.  Re-configuring (with nothing carved) gives the arena back its defaults."""
    import ctypes as C
    import gc
    import ginsim
    from ginsim import _lib
    gc.collect()                        # the Sims of the test before this one hold regions of the device's arena until they are collected
    ctx.sync()
    ctx.release_pool()
    assert ctx.placed_info()['mapped_bytes'] == 0, ctx.placed_info()
    tiny = _lib.PlacedOptions(stripe_bytes=512 << 20, budget_bytes=3 * (512 << 20), limit_bytes=0, search_seconds=0.0)
    assert ginsim.lib.ginsim_placed_configure(ctx.handle, C.byref(tiny)) == 0
    other = ginsim.Context(0)           # a fresh context: its own enabled / note state, the device's arena
    try:
        assert other.placed_reserve(2 * G) is False
        assert not other.placed_enabled and 'budget' in other.placed_note, other.placed_note
        job = _job(other, runs=8192, n=1000, algos=('free',), keep_sensors=True, keep_traj=True, placed=True).run()        # asks, gets plain planes
        assert job.placement()['placed'] == [] and job.placement()['arena']['available'] == 0
        ref = _job(ctx, runs=8192, n=1000, algos=('free',), keep_sensors=True, keep_traj=True, placed=False).run()
        np.testing.assert_array_equal(job.end_errors('free'), ref.end_errors('free'))
        job.release(); ref.release()
    finally:
        other.close()
    # a limit of one stripe: the arena exists but cannot take a 2 GiB request
    small = _lib.PlacedOptions(stripe_bytes=0, budget_bytes=0, limit_bytes=512 << 20, search_seconds=0.0)
    assert ginsim.lib.ginsim_placed_configure(ctx.handle, C.byref(small)) == 0
    third = ginsim.Context(0)
    try:
        assert third.placed_reserve(2 * G) is False and 'limit' in third.placed_note, third.placed_note
    finally:
        third.close()
    # defaults again
    assert ginsim.lib.ginsim_placed_configure(ctx.handle, C.byref(_lib.PlacedOptions())) == 0
    assert ctx.placed_reserve(1 * G), ctx.placed_note
    assert ctx.placed_info()['available'] == 1
    # configuring while the arena exists is refused
    with pytest.raises(ValueError, match='exists already'):
        _lib.check(ginsim.lib.ginsim_placed_configure(ctx.handle, C.byref(tiny)))
