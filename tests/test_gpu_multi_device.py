"""Several GPUs from ONE process (ginsim.multi, Sim(devices=...)) and the thread re-entrancy of the C ABI (SURVEY 8(b):
"re-entrant per context handle ... N Python threads may drive N handles").

The reference's Monte-Carlo loop being sharded is gnss_ins_sim/sim/ins_sim.py:490-506.  The GPU box has ONE device, so the
contexts of these tests share GPU 0 (``devices=[0, 0, 0, 0]``: four contexts, four streams, four host threads); the tests at
the end switch on by themselves when two or more devices are visible and run the same checks on DISTINCT devices, plus real
RCCL with more than one rank."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import REPO, PKG

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        import ginsim
        return ginsim.device_count()
    except Exception:
        return 0


NDEV = _ndev()
need2 = pytest.mark.skipif(NDEV < 2, reason='needs >= 2 visible HIP devices, this box has %d' % NDEV)


def _workload(rf=1):
    import ginsim
    from ginsim import workloads
    fs = 100.0
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', fs, rf)
    acc, gyr = workloads.imu_grade('mid-accuracy')
    return ginsim, fs, rf, ini, truth, acc, gyr


def _check_jobset_against_one_launch(devices, runs=1003):
    """Per-run results bit-identical to ONE launch on one context; merged statistics equal to its on-device reduction."""
    ginsim, fs, rf, ini, truth, acc, gyr = _workload()
    from ginsim import multi
    odo_err = {'scale': 0.999, 'stdv': 0.1}
    kw = dict(algos=('free', 'odo'), odo_err=odo_err, seed=20240917, keep_sensors=True, keep_traj=True)
    ctx = ginsim.Context(devices[0])
    one = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=runs, run_offset=5, ini_first=0, **kw).run()
    ds = multi.DeviceSet(devices)
    js = multi.JobSet(ds, fs, rf, truth, acc, gyr, ini, runs, run_offset=5, ini_first=0, **kw).run()
    assert sum(c for _, c in js.ranges) == runs and len(js.parts) == len(devices)
    # runs either side of every shard boundary, out of order
    edge = sorted({0, runs - 1} | {f for f, c in js.ranges if c} | {max(f - 1, 0) for f, c in js.ranges if c})
    ids = np.array(edge[::-1] + [runs // 2], dtype=np.int64)
    for algo in ('free', 'odo'):
        np.testing.assert_array_equal(js.end_errors(algo), one.end_errors(algo))
        for a, b in zip(js.trajectories(algo, ids), one.trajectories(algo, ids)):
            np.testing.assert_array_equal(a, b)
        m, s = js.stats(algo), one.stats(algo)
        assert m.count == s.count == runs
        np.testing.assert_array_equal(m.maxabs, s.maxabs)
        np.testing.assert_allclose(m.mean, s.mean, rtol=1e-11, atol=1e-18)
        np.testing.assert_allclose(m.std, s.std, rtol=1e-11)
        # and against NumPy on the per-run errors (np.std(ddof=0), ins_data_manager.py:808)
        e = js.end_errors(algo)
        np.testing.assert_allclose(m.std, e.std(0), rtol=1e-10)
    for name in ('accel', 'gyro', 'odo'):
        np.testing.assert_array_equal(js.sensors(name, ids), one.sensors(name, ids))
    js.release()
    one.release()
    ds.close()
    ctx.close()


def test_four_contexts_four_threads_one_gpu():
    _check_jobset_against_one_launch([0, 0, 0, 0])


def test_more_contexts_than_runs():
    """Three runs over four contexts: one context holds nothing, the merge skips its empty record."""
    _check_jobset_against_one_launch([0, 0, 0, 0], runs=3)


def test_raw_threads_on_the_c_abi():
    """No DeviceSet: four plain Python threads, each with its own context handle, at the same time -- create, upload, launch,
    reduce, destroy.  A fifth thread makes the library fail on purpose meanwhile: its error string must stay its own
    (thread-local), the others must neither see it nor be disturbed."""
    ginsim, fs, rf, ini, truth, acc, gyr = _workload()
    import ctypes as C
    from ginsim._lib import lib
    ref_ctx = ginsim.Context(0)
    whole = ginsim.MonteCarloJob(ref_ctx, fs, rf, truth, acc, gyr, ini, runs=2048, seed=99).run()
    ref = whole.end_errors('free')
    whole.release()
    out, errs, start = {}, [], threading.Barrier(5)

    def worker(k):
        try:
            start.wait()
            ctx = ginsim.Context(0)
            for rep in range(3):
                job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=512, run_offset=512 * k, ini_first=512 * k,
                                           seed=99).run()
                out[k] = (job.end_errors('free'), job.stats('free'), lib.ginsim_last_error().decode())
                job.release()
            ctx.close()
        except Exception as e:          # noqa: BLE001
            errs.append(repr(e))

    def saboteur():
        start.wait()
        for _ in range(200):
            rc = lib.ginsim_mc_run(ref_ctx.handle, None)
            assert rc != 0 and b'NULL' in lib.ginsim_last_error()
        out['bad'] = lib.ginsim_last_error().decode()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)] + [threading.Thread(target=saboteur)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    assert 'NULL' in out['bad']
    for k in range(4):
        e, st, msg = out[k]
        np.testing.assert_array_equal(e, ref[512 * k:512 * (k + 1)])
        assert st.count == 512 and 'NULL' not in msg
    from ginsim import StatsResult
    merged = StatsResult.merge([out[k][1].pack() for k in range(4)])
    np.testing.assert_allclose(merged.std, ref.std(0), rtol=1e-10)
    np.testing.assert_array_equal(merged.maxabs, np.abs(ref).max(0))
    ref_ctx.close()


def _sim(devices=None, runs=300, keep='auto', rf=1, env_devices=None, **kw):
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration, free_integration_odo
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=9, gps=True, odo=True, odo_opt={'scale': 0.999, 'stdv': 0.1})
    algos = [free_integration_odo.FreeIntegration(ini), free_integration.FreeIntegration(ini)]
    old = os.environ.pop('GINSIM_DEVICES', None)
    if env_devices is not None:
        os.environ['GINSIM_DEVICES'] = env_devices
    try:
        sim = ins_sim.Sim([100.0, 10.0, 100.0], csv, ref_frame=rf, imu=imu, algorithm=algos, seed=4242, devices=devices,
                          keep_trajectories=keep, geo_mag_n=[30.0, -3.0, 40.0], **kw)
    finally:
        os.environ.pop('GINSIM_DEVICES', None)
        if old is not None:
            os.environ['GINSIM_DEVICES'] = old
    sim.run(runs)
    return sim


def _compare_sims(a, b, runs, tmp_path, capsys):
    for s, extra in ((-1, ''), (0, '')):
        a.results(err_stats_start=s, extra_opt=extra)
        b.results(err_stats_start=s, extra_opt=extra)
        capsys.readouterr()
        for dn in ('att_euler', 'pos', 'vel'):
            for stat in ('max', 'avg', 'std'):
                x, y = a.err_stats[dn][stat], b.err_stats[dn][stat]
                if hasattr(x, 'keys'):
                    assert sorted(x.keys()) == sorted(y.keys())
                    for k in list(x.keys())[::37]:       # per-run process statistics: the reduction over time groups 64 runs
                        np.testing.assert_allclose(x[k], y[k], rtol=1e-11, atol=1e-18)     # of a launch (Chan merges in LDS)
                else:
                    np.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-15)
    if 'accel' in a.dmgr.available and len(a.dmgr.accel.data):
        assert list(a.dmgr.accel.data.keys()) == list(b.dmgr.accel.data.keys())
        for r in (0, runs // 2 - 1, runs // 2, runs - 1):
            for name in ('accel', 'gyro', 'odo', 'gps', 'mag'):
                np.testing.assert_array_equal(getattr(a.dmgr, name).data[r], getattr(b.dmgr, name).data[r])
            for name in ('pos', 'vel', 'att_euler', 'att_quat'):
                for algo in ('algo0', 'algo1'):
                    np.testing.assert_array_equal(getattr(a.dmgr, name).data['%s_%d' % (algo, r)],
                                                  getattr(b.dmgr, name).data['%s_%d' % (algo, r)])


def test_sim_devices_equals_sim_on_one_context(tmp_path, capsys):
    """Sim(devices=[0, 0]) on the one-GPU box: the same per-run series, the same end-point and process statistics as the plain
    Sim -- everything kept (two algorithms, 9-axis IMU + GPS + odometer)."""
    one = _sim(None)
    two = _sim([0, 0])
    assert two.mc.devices == [0, 0] and one.mc.devices is None
    _compare_sims(two, one, 300, tmp_path, capsys)


def test_sim_devices_with_a_psd_environment(capsys):
    """Sim(env=<PSD arrays>, devices=[0, 0]) (ABI 8): every context makes the vibration series of ITS runs (phases by the global run
    id, an FFT plan of its own per stream) -- the same sensors and trajectories as the plain Sim to rounding (the series of a run
    come out of FFT batches of other sizes), the gyroscope's PSD given on the series' grid and therefore halved per global run."""
    n = 1000
    f = np.linspace(0.0, 50.0, n // 2 + 1)
    def env():
        g = 1e-6 * (1.0 + np.cos(f / 7.0) ** 2)
        return {'acc': np.array([[0.0, 1e-3, 2e-3, 1e-3], [12.0, 4e-3, 1e-3, 2e-3], [45.0, 1e-3, 1e-3, 1e-3]]), 'gyro': np.stack([f, g, 0.5 * g, 2.0 * g], axis=1)}
    e1, e2 = env(), env()
    one = _sim(None, runs=200, keep=True, env=e1)
    two = _sim([0, 0], runs=200, keep=True, env=e2)
    assert two.mc.devices == [0, 0]
    np.testing.assert_array_equal(e1['gyro'], e2['gyro'])              # both left the caller's array as the reference does: halved 200 times
    assert e1['gyro'][5, 1] == 0.0 or e1['gyro'][5, 1] < 1e-60 and e1['gyro'][0, 1] == env()['gyro'][0, 1]
    for r in (0, 99, 100, 199):
        for name, tol in (('accel', 1e-13), ('gyro', 1e-15), ('odo', 0.0)):
            np.testing.assert_allclose(getattr(two.dmgr, name).data[r], getattr(one.dmgr, name).data[r], rtol=0, atol=tol)
        for name, tol in (('vel', 1e-10), ('att_euler', 1e-11)):
            for algo in ('algo0', 'algo1'):
                np.testing.assert_allclose(getattr(two.dmgr, name).data['%s_%d' % (algo, r)], getattr(one.dmgr, name).data['%s_%d' % (algo, r)], rtol=0, atol=tol)
    plain = _sim(None, runs=200, keep=True)
    assert np.abs(one.dmgr.accel.data[3] - plain.dmgr.accel.data[3]).std() > 0.05        # the environment is there
    capsys.readouterr()


def test_sim_devices_writes_the_same_files(tmp_path, capsys):
    """results(data_dir) of a Sim spread over two contexts: the CSV files of the saved runs and the summary are byte-identical
    to the single-context Sim's (Sim_data.save_to_file, sim_data.py:117-165; the views route every run to the device that holds it)."""
    import filecmp
    one, two = _sim(None, runs=40), _sim([0, 0], runs=40)
    da, db = tmp_path / 'one', tmp_path / 'two'
    one.results(str(da), err_stats_start=-1, max_saved_runs=40)
    two.results(str(db), err_stats_start=-1, max_saved_runs=40)
    capsys.readouterr()
    files = sorted(os.listdir(da))
    assert files == sorted(os.listdir(db)) and len(files) > 200 and 'accel-39.csv' in files and 'pos-algo1_21.csv' in files
    for f in files:
        if f == 'summary.txt':          # the two summaries name their own directories
            a, b = open(da / f).read().replace(str(da), ''), open(db / f).read().replace(str(db), '')
            assert a == b
        else:
            assert filecmp.cmp(da / f, db / f, shallow=False), f


def test_sim_devices_statistics_only(tmp_path, capsys):
    """The statistics-only launches (online process statistics, NED end-point record, a few kept runs) spread the same way."""
    one = _sim(None, runs=1001, keep=False, rf=0, keep_runs=5)
    three = _sim([0, 0, 0], runs=1001, keep=False, rf=0, keep_runs=5)
    _compare_sims(three, one, 5, tmp_path, capsys)
    for s in (one, three):
        s.results(err_stats_start=-1, extra_opt='ned')
    capsys.readouterr()
    for grp in ('algo0', 'algo1'):                       # two algorithms: one group of end-point statistics each
        np.testing.assert_allclose(three.err_stats['pos']['std'][grp], one.err_stats['pos']['std'][grp], rtol=1e-10)
        np.testing.assert_array_equal(three.err_stats['pos']['max'][grp], one.err_stats['pos']['max'][grp])
        assert np.all(one.err_stats['pos']['max'][grp] < 50.0)       # metres, not radians


def test_sim_devices_from_the_environment(capsys):
    """An UNCHANGED script takes the device list from $GINSIM_DEVICES."""
    sim = _sim(None, runs=64, env_devices='0,0')
    assert sim.mc.devices == [0, 0]
    sim = _sim(None, runs=64, env_devices='all')
    assert sim.mc.devices == list(range(NDEV))


def test_sim_devices_refuses_what_it_cannot_do():
    from gnss_ins_sim.sim import imu_model, ins_sim
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    with pytest.raises(ValueError, match='out of range'):
        ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, devices=[0, 99]).run(4)
    with pytest.raises(ValueError, match='not both'):
        ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, devices=[0], device=0).run(4)
    with pytest.raises(ValueError):
        ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, devices='some').run(4)


def test_allan_plugin_over_a_device_set(capsys):
    """The device-resident Allan plugin (demo_algorithms.allan_analysis.run_device) takes a JobSet: every device analyses the
    series it generated; per-run results equal the single-context Sim's."""
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import allan_analysis
    csv = os.path.join(PKG, 'motion_profiles', 'static_1800s.csv')
    imu = imu_model.IMU(accuracy='low-accuracy', axis=6, gps=False)
    res = []
    for dev in (None, [0, 0]):
        sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=allan_analysis.Allan(), seed=5, devices=dev)
        sim.run(4)
        res.append(sim)
    for r in range(4):
        for name in ('ad_accel', 'ad_gyro', 'algo_time'):
            np.testing.assert_array_equal(res[1].dmgr.get_data_all(name).data['algo0_%d' % r],
                                          res[0].dmgr.get_data_all(name).data['algo0_%d' % r])


# ------------------------------------------------------------------------------------------------------------------------
# Switched on by the hardware: two or more visible devices (the first multi-GPU box must not execute untested code paths)
@need2
def test_distinct_devices_bit_identical_to_one_launch():
    _check_jobset_against_one_launch(list(range(NDEV)), runs=4099)


@need2
def test_sim_on_all_devices(tmp_path, capsys):
    one = _sim(None, runs=2000)
    every = _sim('all', runs=2000)
    assert every.mc.devices == list(range(NDEV))
    _compare_sims(every, one, 2000, tmp_path, capsys)


@need2
def test_wave_specialised_kernels_launch_on_every_device():
    """The 100 KB dynamic-LDS launches need their function attribute on EVERY device (csrc/device_once.hpp): C2-shaped fp64
    and fp32 batches on the last device after the first one has already run them."""
    ginsim, fs, rf, ini, truth, acc, gyr = _workload()
    ends = []
    for d in (0, NDEV - 1):
        ctx = ginsim.Context(d)
        for prec in ('f64', 'f32'):
            job = ginsim.MonteCarloJob(ctx, fs, rf, truth, acc, gyr, ini, runs=4096, seed=3, precision=prec, keep_traj=True).run()
            assert 'split' in job.kernel_name()
            ends.append(job.end_errors('free'))
            job.release()
        ctx.close()
    np.testing.assert_array_equal(ends[0], ends[2])
    np.testing.assert_array_equal(ends[1], ends[3])


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@need2
def test_bench_real_rccl_world_n():
    """`python bench.py --gpus N` as the driver runs it: one rank per GPU, backend nccl (= RCCL over xGMI), the library's own
    all-gather behind the C ABI -- no --shared-device, no gloo."""
    n = 2 if NDEV < 4 else (4 if NDEV < 8 else 8)
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1', '--exchange', 'abi',
           '--pmc', 'off', '--no-legs']
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == n and line['scaling'] == 'weak'
    assert line['config']['total_runs_per_step'] == n * line['config']['runs_per_gpu']
    assert line['result']['runs'] == line['config']['total_runs_per_step']
    assert 'RCCL all-gather behind the C ABI' in line['config']['parallelism'], line['config']['parallelism']


_NCCL_SIM = r'''
import os, sys, json
import numpy as np
rank = int(sys.argv[1]); world = int(sys.argv[2])
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT='%(port)d')
sys.path[:0] = [%(pkg)r, %(repo)r]
import torch, torch.distributed as dist
torch.cuda.set_device(rank)
dist.init_process_group('nccl', rank=rank, world_size=world)
from gnss_ins_sim.sim import imu_model, ins_sim
from demo_algorithms import free_integration
csv = os.path.join(%(pkg)r, 'motion_profiles', 'turn_90deg.csv')
ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
ini[0:2] *= np.pi / 180; ini[6:9] *= np.pi / 180
sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False),
                  algorithm=free_integration.FreeIntegration(ini), seed=77)
sim.run(1001)
sim.results(err_stats_start=-1)
if rank == 0:
    print('RESULT ' + json.dumps({'vel_std': [float(x) for x in sim.err_stats['vel']['std']],
                                  'att_max': [float(x) for x in sim.err_stats['att_euler']['max']], 'exchange': sim.mc.exchange}))
dist.barrier(); dist.destroy_process_group()
'''


@need2
def test_sim_under_real_nccl(tmp_path):
    """Sim under torch.distributed with backend nccl and one rank per device: the merged statistics equal the single-process
    Sim's over the same 1001 runs; the exchange is the library's RCCL all-gather."""
    world = min(NDEV, 4)
    script = tmp_path / 'n.py'
    script.write_text(_NCCL_SIM % {'pkg': PKG, 'repo': REPO, 'port': _port()})
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    got = json.loads([l for l in outs[0].splitlines() if l.startswith('RESULT ')][-1][7:])
    assert got['exchange'] == 'abi', got
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False),
                      algorithm=free_integration.FreeIntegration(ini), seed=77)
    sim.run(1001)
    sim.results(err_stats_start=-1)
    np.testing.assert_allclose(got['vel_std'], sim.err_stats['vel']['std'], rtol=1e-10)
    np.testing.assert_array_equal(got['att_max'], sim.err_stats['att_euler']['max'])
