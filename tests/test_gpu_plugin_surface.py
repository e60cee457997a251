"""The plugin surface the way the reference uses it (VERDICT r01 "next" #2):

* a USER plugin (no ``mc_algo``: plain ``.input/.output/.run/.get_results/.reset`` over host arrays) hosted by this
  package's ``Sim.run(R)`` -- ``InsAlgoMgr.run_algo``'s per-run ``reset() -> run(deepcopy(inputs)) -> get_results()`` loop
  (/root/reference/gnss_ins_sim/sim/ins_algo_manager.py:39-96), keys '<algo>_<run>' -- next to the fused FreeIntegration:
  same sensors, so the two outputs must agree per sample, and both get error statistics;
* ``Sim(algorithm=None)`` / hosted-only Sims save their data (save_data without a fused plugin);
* logged-data ingestion: ``results(data_dir)`` then ``Sim(motion_def=<dir>)`` re-reads the CSV files and integrates them
  through the plugin's ``run(set_of_input)`` (/root/reference/gnss_ins_sim/sim/ins_sim.py:426-442,
  /root/reference/demo_gen_data_from_files.py:24-80).
"""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import PKG, ang_close

pytestmark = pytest.mark.gpu
D2R = np.pi / 180
TURN = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')


def _ini():
    ini = np.genfromtxt(TURN, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= D2R
    ini[6:9] *= D2R
    return ini


class HostedFreeIntegration(object):
    """A user's plugin: the NumPy restatement of free integration behind the reference's duck-typed surface.
    It is NOT inside the fused kernel (no mc_algo), so the host must loop over runs and hand it host arrays."""

    def __init__(self, ini, name=None):
        self.input = ['ref_frame', 'fs', 'gyro', 'accel']
        self.output = ['att_euler', 'pos', 'vel']
        self.ini = np.array(ini, dtype=np.float64)
        self.results = None
        self.calls, self.resets = [], 0
        if name:
            self.name = name

    def run(self, set_of_input):
        from oracle import ins_np
        rf, fs, gyro, accel = set_of_input
        assert isinstance(gyro, np.ndarray) and gyro.ndim == 2 and gyro.flags['WRITEABLE']
        self.calls.append(gyro[:3].copy())
        att, pos, vel = ins_np.free_integration(int(rf), float(fs), gyro[None], accel[None], self.ini)
        gyro[:] = 0.0                                  # the host hands out deep copies: trashing them must be harmless
        self.results = [att[0], pos[0], vel[0]]

    def get_results(self):
        return self.results

    def reset(self):
        self.resets += 1


@pytest.mark.parametrize('rf', [0, 1])
def test_hosted_plugin_next_to_the_fused_one(rf):
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    ini = _ini()
    fused, hosted = free_integration.FreeIntegration(ini), HostedFreeIntegration(ini, name='user')
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    sim = ins_sim.Sim([100.0, 0.0, 0.0], TURN, ref_frame=rf, imu=imu, algorithm=[fused, hosted], seed=11)
    R = 5
    sim.run(R)
    assert hosted.resets == R and len(hosted.calls) == R
    d = sim.dmgr
    keys = list(d.att_euler.data.keys())
    assert keys == ['algo0_%d' % r for r in range(R)] + ['user_%d' % r for r in range(R)]
    for r in range(R):
        # the plugin saw run r's sensors (a deep copy of them)
        np.testing.assert_array_equal(hosted.calls[r], d.gyro.data[r][:3])
        assert np.any(d.gyro.data[r] != 0.0)
        assert ang_close(d.att_euler.data['user_%d' % r], d.att_euler.data['algo0_%d' % r], 1e-9)
        np.testing.assert_allclose(d.vel.data['user_%d' % r], d.vel.data['algo0_%d' % r], rtol=0, atol=1e-9)
        np.testing.assert_allclose(d.pos.data['user_%d' % r], d.pos.data['algo0_%d' % r], rtol=0,
                                   atol=2e-8 if rf == 1 else 1e-12)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        sim.results(err_stats_start=-1)
    text = out.getvalue()
    for name in ('att_euler', 'pos', 'vel'):
        st = sim.err_stats[name]
        assert set(st['max'].keys()) == {'algo0', 'user'}, st['max'].keys()       # grouped like ins_data_manager.py:810-832
        for s in ('max', 'avg', 'std'):
            tol = dict(rtol=1e-6, atol=1e-9) if name != 'pos' or rf == 1 else dict(rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(st[s]['user'], st[s]['algo0'], **tol)
    assert 'statistics for simulation velocity from algo' in text
    # process statistics (the reference's default err_stats_start=0): per-run dicts for both plugins
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(err_stats_start=2.0)
    st = sim.err_stats['vel']
    assert set(st['std'].keys()) == set(keys)
    for r in range(R):
        np.testing.assert_allclose(st['std']['user_%d' % r], st['std']['algo0_%d' % r], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(st['max']['user_%d' % r], st['max']['algo0_%d' % r], rtol=1e-6, atol=1e-10)


def test_hosted_only_and_no_algorithm_save_their_data(tmp_path):
    """results(data_dir) without a fused plugin (demo_no_algo.py / demo_multiple_algorithms.py call it)."""
    from gnss_ins_sim.sim import imu_model, ins_sim
    imu = imu_model.IMU(accuracy='low-accuracy', axis=6, gps=True)
    sim = ins_sim.Sim([100.0, 10.0, 0.0], TURN, ref_frame=0, imu=imu, algorithm=None, seed=3)
    sim.run(4)
    with contextlib.redirect_stdout(io.StringIO()):
        avail = sim.results(str(tmp_path / 'none'))
    files = sorted(os.listdir(tmp_path / 'none'))
    assert 'accel-3.csv' in files and 'gyro-0.csv' in files and 'gps-2.csv' in files and 'ref_pos.csv' in files
    assert 'accel' in avail
    # hosted-only: statistics come from the host computation of the reference
    hosted = HostedFreeIntegration(_ini())
    sim = ins_sim.Sim([100.0, 0.0, 0.0], TURN, ref_frame=1, imu=imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False),
                      algorithm=hosted, seed=3)
    sim.run(3)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(str(tmp_path / 'hosted'), err_stats_start=-1)
    files = sorted(os.listdir(tmp_path / 'hosted'))
    assert 'vel-algo0_2.csv' in files and 'att_euler-algo0_0.csv' in files and 'summary.txt' in files
    e = np.stack([sim.dmgr.vel.data['algo0_%d' % r][-1] - sim.dmgr.ref_vel.data[-1] for r in range(3)])
    np.testing.assert_allclose(sim.err_stats['vel']['std'], e.std(0), rtol=1e-12)
    np.testing.assert_allclose(sim.err_stats['vel']['max'], np.abs(e).max(0), rtol=1e-12)


def test_two_fused_algorithms_save_both(tmp_path):
    """with two fused algorithms and more runs than max_saved_runs, the files of BOTH are written."""
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration, free_integration_odo
    ini = _ini()
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False, odo=True, odo_opt={'scale': 0.999, 'stdv': 0.1})
    sim = ins_sim.Sim([100.0, 0.0, 0.0], TURN, ref_frame=1, imu=imu,
                      algorithm=[free_integration.FreeIntegration(ini), free_integration_odo.FreeIntegration(ini)], seed=5)
    sim.run(100)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(str(tmp_path), err_stats_start=-1, max_saved_runs=3)
    files = set(os.listdir(tmp_path))
    for a in ('algo0', 'algo1'):
        for r in range(3):
            assert 'pos-%s_%d.csv' % (a, r) in files
        assert 'pos-%s_3.csv' % a not in files
    assert 'accel-2.csv' in files and 'accel-3.csv' not in files


@pytest.mark.parametrize('rf', [0, 1])
def test_logged_data_round_trip(tmp_path, rf):
    """Save a Monte-Carlo Sim with results(data_dir), re-ingest the directory, integrate the logged sensors with the
    plugin's run(set_of_input): the trajectories must be those of the fused kernel that generated the files."""
    from gnss_ins_sim.sim import imu_model, ins_sim
    from demo_algorithms import free_integration
    ini = _ini()
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    R = 3
    gen = ins_sim.Sim([100.0, 0.0, 0.0], TURN, ref_frame=rf, imu=imu, algorithm=free_integration.FreeIntegration(ini), seed=21)
    gen.run(R)
    with contextlib.redirect_stdout(io.StringIO()):
        gen.results(str(tmp_path), err_stats_start=-1)
    algo = free_integration.FreeIntegration(ini)
    sim = ins_sim.Sim([100.0, 0.0, 0.0], str(tmp_path), ref_frame=rf, imu=None, algorithm=algo)
    sim.run(R)
    assert algo.run_times == R
    d = sim.dmgr
    assert sorted(d.gyro.data.keys()) == list(range(R))
    for r in range(R):
        np.testing.assert_allclose(d.gyro.data[r], gen.dmgr.gyro.data[r], rtol=0, atol=1e-16)     # deg/s in the file -> rad/s
        key = 'algo0_%d' % r
        assert ang_close(d.att_euler.data[key], gen.dmgr.att_euler.data[key], 1e-11)
        np.testing.assert_allclose(d.vel.data[key], gen.dmgr.vel.data[key], rtol=0, atol=1e-11)
        np.testing.assert_allclose(d.pos.data[key], gen.dmgr.pos.data[key], rtol=0, atol=2e-8 if rf == 1 else 1e-13)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(err_stats_start=-1)
    # end-point statistics of the re-ingested run == those of the generating run (host computation vs device reduction)
    for name in ('att_euler', 'vel'):
        np.testing.assert_allclose(sim.err_stats[name]['std'], gen.err_stats[name]['std'], rtol=1e-6, atol=1e-10)


def test_integration_md_binding_runs_as_written():
    """INTEGRATION.md section B1 -- the ctypes binding a maintainer of the reference would add -- executed as written
    (only the library path is made absolute; `demo_algorithms.free_integration` resolves to the drop-in class, whose
    constructor leaves the same attributes as the reference's, free_integration.py:19-61), on the reference's own
    bosch fixture: external gravity + a two-column table of initial states, and the hosting contract of run()."""
    import re
    import types
    from conftest import load_golden, assert_traj_close, REPO
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    block = re.search(r'### B1\..*?```python\n(.*?)```', text, re.S).group(1)
    lib = os.path.join(PKG, 'lib', 'libginsim.so')
    assert "'libginsim.so'" in block
    mod = types.ModuleType('free_integration_hip')
    exec(compile(block.replace("'libginsim.so'", repr(lib)), 'INTEGRATION.md#B1', 'exec'), mod.__dict__)
    g = load_golden('t1_fixture_bosch')
    k = g['rows']
    for tag, rf, ini, erot in (('extg', 0, g['ini'], False), ('wgs', 0, g['ini'][:9], True), ('rf1', 1, g['ini'][:9], True)):
        algo = mod.FreeIntegration(ini.copy(), earth_rot=erot)
        assert algo.input == ['ref_frame', 'fs', 'gyro', 'accel'] and algo.output == ['att_euler', 'pos', 'vel']
        algo.run([rf, float(g['fs']), g['gyro'].copy(), g['accel'].copy()])
        att, pos, vel = algo.get_results()
        assert_traj_close(att[k], pos[k], vel[k], g['att_' + tag], g['pos_' + tag], g['vel_' + tag], rtol=1e-10, what='B1 ' + tag)
    # two sets of initial states: the second run() uses the second column (free_integration.py:76-88)
    two = np.stack([g['ini'][:9], g['ini'][:9]], axis=1)
    two[2, 1] += 100.0
    algo = mod.FreeIntegration(two)
    algo.run([1, float(g['fs']), g['gyro'], g['accel']])
    first = [a.copy() for a in algo.get_results()]
    algo.run([1, float(g['fs']), g['gyro'], g['accel']])
    assert abs(algo.get_results()[1][0, 2] - first[1][0, 2]) > 1.0 or abs(np.linalg.norm(algo.get_results()[1][0] - first[1][0])) > 1.0
    with pytest.raises(ValueError, match='bad sizes'):       # error convention: status code + ginsim_last_error -> ValueError
        algo.run([1, float(g['fs']), np.empty((0, 3)), np.empty((0, 3))])
