"""CPU-side tests: the C-ABI library loads and exports every symbol include/ginsim.h declares, host-side
argument checking, native pathgen vs the reference goldens, sharding + the all-reduce path on gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, REPO, PKG


def test_library_exports_every_declared_symbol():
    import ctypes
    import ginsim
    hdr = open(os.path.join(REPO, 'include', 'ginsim.h')).read()
    declared = set(re.findall(r'\b(ginsim_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 20
    raw = ctypes.CDLL(ginsim.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, 'declared in include/ginsim.h but not exported: %s' % missing
    assert set(ginsim.EXPORTS) <= declared
    assert ginsim.lib.ginsim_abi_version() == 8


def test_header_is_plain_c(tmp_path):
    """include/ginsim.h is the C ABI: it must compile as C99 on its own (what a cgo / JNI / ctypes-generator user feeds it to)."""
    import subprocess
    src = tmp_path / 'h.c'
    src.write_text('#include "ginsim.h"\nint main(void) { return GINSIM_ABI_VERSION == 8 ? 0 : 1; }\n')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I' + os.path.join(REPO, 'include'), '-fsyntax-only', str(src)],
                   check=True, timeout=120)


def test_no_gpu_fails_loudly():
    import ginsim
    if ginsim.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(ginsim.GinsimError, match='no CPU fallback'):
        ginsim.Context(0)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert 'ginsim_oracle' not in src, f


@pytest.mark.parametrize('rf', [0, 1])
def test_native_pathgen_matches_reference(rf):
    import ginsim
    g = load_golden('t2_turn_rf%d' % rf)
    k = g['rows']
    r = ginsim.pathgen(g['ini_pva'], g['motion_def'], 100.0, 10.0, g['mobility'], rf, gps=True)
    assert r['imu'].shape == (int(g['n']), 7)
    np.testing.assert_allclose(r['imu'][:, 1:4], g['full_ref_accel'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['imu'][:, 4:7], g['full_ref_gyro'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(r['nav'][k, 1:4], g['ref_pos'], rtol=1e-15, atol=0)
    np.testing.assert_allclose(r['nav'][k, 4:7], g['ref_vel'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['nav'][k, 7:10], g['ref_att'], rtol=0, atol=1e-14)
    np.testing.assert_allclose(r['gps'][:, 1:7], g['ref_gps'], rtol=1e-15, atol=1e-13)
    np.testing.assert_allclose(r['gps'][:, 7], g['gps_vis'])
    np.testing.assert_allclose(r['odo'][k, 2], g['ref_odo'], rtol=0, atol=1e-13)


@pytest.mark.parametrize('rf', [0, 1])
def test_native_pathgen_every_command_type_and_custom_mode(rf):
    """Command types 1-5 (type 4 = absolute attitude + relative velocity appears in none of the reference's own motion
    files) and a custom mobility array (Sim(mode=np.array([...])), ins_sim.py:612-640) against the executed reference."""
    import ginsim
    from gnss_ins_sim.sim import ins_sim
    g = load_golden('truth_mixed_types_rf%d' % rf)
    k, kg = g['rows'], g['gps_rows']
    ini, md = __import__('ginsim.workloads', fromlist=['parse_motion']).parse_motion(str(g['text']))
    np.testing.assert_allclose(ini, g['ini_pva'], rtol=0, atol=0)
    np.testing.assert_allclose(md, g['motion_def'], rtol=0, atol=0)
    assert sorted(set(md[:, 0])) == [1.0, 2.0, 3.0, 4.0, 5.0]
    mob = ins_sim.Sim._parse_mode(g['mode'])
    np.testing.assert_allclose(mob, g['mobility'], rtol=0, atol=0)
    r = ginsim.pathgen(ini, md, float(g['fs']), float(g['fs_gps']), mob, rf, gps=True)
    assert r['imu'].shape[0] == int(g['n']) and r['gps'].shape[0] == int(g['m'])
    np.testing.assert_allclose(r['imu'][k], g['imu'], rtol=0, atol=2e-13)
    np.testing.assert_allclose(r['nav'][k, 0:4], g['nav'][:, 0:4], rtol=1e-15, atol=0)
    np.testing.assert_allclose(r['nav'][k, 4:10], g['nav'][:, 4:10], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['gps'][kg], g['gps'], rtol=1e-15, atol=1e-13)
    np.testing.assert_allclose(r['odo'][k], g['odo'], rtol=1e-14, atol=1e-13)
    with pytest.raises(TypeError):
        ins_sim.Sim._parse_mode(np.array([1.0, 2.0]))


def test_pathgen_error_behaviour():
    """Same exception type and wording as pathgen.py:117-125 for bad motion definitions."""
    import ginsim
    g = load_golden('t2_turn_rf1')
    md = g['motion_def'].copy()
    md[1, 7] = -1.0
    with pytest.raises(ValueError, match='negative time duration'):
        ginsim.pathgen(g['ini_pva'], md, 100.0, 0.0, g['mobility'], 1)
    md = g['motion_def'].copy()
    md[:, 7] = 0.0
    with pytest.raises(ValueError, match='must be above 0'):
        ginsim.pathgen(g['ini_pva'], md, 100.0, 0.0, g['mobility'], 1)
    md = g['motion_def'].copy()
    md[0, 0] = 7
    with pytest.raises(ValueError, match='unsupported motion type'):
        ginsim.pathgen(g['ini_pva'], md, 100.0, 0.0, g['mobility'], 1)


def test_sensor_model_coefficients():
    import ginsim
    from oracle import ins_np
    err = {'b': np.array([1e-3, 0, -1e-3]), 'b_drift': np.array([1e-4, 2e-4, 3e-4]),
           'b_corr': np.array([100.0, np.inf, 50.0]), 'vrw': np.array([0.01, 0.02, 0.03])}
    m = ginsim.sensor_model(err, 'vrw', 200.0)
    a, b, white = ins_np.gm_coeffs(err['b_corr'], err['b_drift'], 200.0)
    np.testing.assert_array_equal(np.array(m.gm_a[:]), a)
    np.testing.assert_allclose(np.array(m.gm_b[:]), b, rtol=1e-15)
    assert list(m.white_drift[:]) == [0, 1, 0]
    np.testing.assert_allclose(np.array(m.white[:]), err['vrw'] / np.sqrt(1 / 200.0), rtol=1e-15)


def test_starts_on_truth_is_what_lets_a_statistics_launch_take_plain_sums():
    """ginsim_mc_params.proc_plain_sums (include/ginsim.h): MonteCarloJob states it only when every initial state lies on the
    truth's first sample -- attitude and position exactly, the velocity to 1e-9 m/s."""
    from ginsim import workloads
    from ginsim.engine import starts_on_truth, ini_table
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 0)          # yaw 315 deg in the table, -45 deg in the truth
    nav0 = np.concatenate([truth['ref_att'][0], truth['ref_pos'][0], truth['ref_vel'][0]])
    table, _ = ini_table(ini)
    assert starts_on_truth(table, nav0) and not starts_on_truth(table, nav0, ref_frame=1)         # ref_frame 1 has no plain form
    for col, d in ((6, 1e-9), (0, 1e-12), (2, 1e-6), (3, 1e-6), (8, -1e-9)):
        off = table.copy()
        off[0, col] += d
        assert not starts_on_truth(off, nav0), col
    two = np.concatenate([table, table])                    # several sets of initial states: every one of them
    assert starts_on_truth(two, nav0)
    two[1, 7] += 1e-6
    assert not starts_on_truth(two, nav0)
    off = table.copy()
    off[0, 3] += 1e-11                                       # below what the velocity check resolves: still on the truth
    assert starts_on_truth(off, nav0)


def test_ini_table_shapes():
    import ginsim
    t, has_g = ginsim.ini_table(np.arange(9.0))
    assert t.shape == (1, 10) and not has_g
    t, has_g = ginsim.ini_table(np.arange(30.0).reshape(10, 3))
    assert t.shape == (3, 10) and has_g and t[1, 9] == 28.0
    with pytest.raises(ValueError):
        ginsim.ini_table(np.zeros((2, 2, 2)))


def test_shard_is_a_partition():
    from ginsim import distributed
    for total, world in ((1048576, 8), (1000, 3), (5, 8), (65536, 1)):
        spans = [distributed.shard(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1


def test_stats_merge_host_matches_numpy():
    import ginsim
    from ginsim import distributed
    rng = np.random.default_rng(3)
    e = rng.normal(size=(1000, 9)) * np.logspace(-6, 3, 9) + np.logspace(-3, 6, 9)
    parts = [distributed.stats_from_errors(x) for x in np.array_split(e, [17, 400, 401])]
    m = ginsim.StatsResult.merge(parts)
    assert m.count == 1000
    np.testing.assert_allclose(m.mean, e.mean(0), rtol=1e-13)
    np.testing.assert_allclose(m.std, e.std(0), rtol=1e-11)
    np.testing.assert_array_equal(m.maxabs, np.abs(e).max(0))


_WORKER = r'''
import os, sys
sys.path[:0] = [%(pkg)r, %(repo)r]
import numpy as np, torch, torch.distributed as dist
import ginsim
from ginsim import distributed
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=%(world)d)
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(11)
e = rng.normal(size=(%(total)d, 9)) * np.logspace(-5, 2, 9) + np.linspace(-1, 1, 9)
first, count = distributed.shard(%(total)d, world, rank)
part = ginsim.StatsResult(ginsim.StatsResult.unpack(distributed.stats_from_errors(e[first:first + count])))
m = distributed.allreduce_stats(part, dist.group.WORLD, torch.device('cpu'))
assert m.count == %(total)d
np.testing.assert_allclose(m.mean, e.mean(0), rtol=1e-12)
np.testing.assert_allclose(m.std, e.std(0), rtol=1e-11)
np.testing.assert_array_equal(m.maxabs, np.abs(e).max(0))
# the non-blocking form bench.py uses (issue now, collect a step later) gives the same record
h = distributed.allreduce_stats_begin(part, dist.group.WORLD, torch.device('cpu'))
m2 = distributed.allreduce_stats_end(h)
np.testing.assert_array_equal(m2.pack(), m.pack())
assert distributed.allreduce_stats_end(distributed.allreduce_stats_begin(part)) is part      # single process: identity
# bootstrap of the library's own communicator: a rank that cannot reach librccl must make EVERY rank raise before any of them
# enters the collective initialisation (nobody is left waiting); with all ranks fine the id of rank 0 reaches every rank
class FakeCtx(object):
    def __init__(self, broken=False, init_fails=False): self.broken, self.init_fails, self.got, self.ids, self.destroyed = broken, init_fails, None, 0, 0
    def comm_probe(self):
        if self.broken: raise OSError('librccl not found')
    def comm_unique_id(self):
        self.ids += 1
        return bytes([rank]) * 128
    def comm_init(self, nranks, r, uid):
        if self.init_fails: raise OSError('ncclCommInitRank: unhandled system error')
        self.got = (nranks, r, uid)
    def comm_destroy(self): self.destroyed += 1
bad = FakeCtx(broken=(rank == 1))
try:
    distributed.init_abi_comm(bad, dist.group.WORLD, torch.device('cpu'))
    raise SystemExit('init_abi_comm did not raise on rank %%d' %% rank)
except RuntimeError as e:
    assert 'not usable on every rank' in str(e) and bad.got is None and bad.ids == 0     # nobody drew an id, nobody entered the init
# the collective initialisation fails on ONE rank: every rank raises, and the rank that did get a communicator drops it again
half = FakeCtx(init_fails=(rank == 1))
try:
    distributed.init_abi_comm(half, dist.group.WORLD, torch.device('cpu'))
    raise SystemExit('init_abi_comm did not raise on rank %%d' %% rank)
except RuntimeError as e:
    assert 'ncclCommInitRank failed' in str(e) and half.destroyed == (1 if rank == 0 else 0)
good = FakeCtx()
assert distributed.init_abi_comm(good, dist.group.WORLD, torch.device('cpu')) == (world, rank)
assert good.got == (world, rank, bytes([0]) * 128) and good.ids == (1 if rank == 0 else 0)      # the id is drawn on rank 0 only
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_two_process_gloo_allreduce_of_stats(tmp_path):
    """world_size 2 over gloo: shard -> per-rank record -> ONE all-reduce -> Chan merge == NumPy on all runs."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'pkg': PKG, 'repo': REPO, 'port': port, 'world': 2, 'total': 1001})
    env = dict(os.environ, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


@pytest.mark.parametrize('rf', [0, 1])
def test_native_pathgen_magnetometer(rf):
    import ginsim
    g = load_golden('t3_mag9_gps_rf%d' % rf)
    t2 = load_golden('t2_turn_rf%d' % rf)
    r = ginsim.pathgen(t2['ini_pva'], t2['motion_def'], 100.0, 10.0, t2['mobility'], rf, gps=True, geo_mag_n=g['geo_mag_n'])
    np.testing.assert_allclose(r['mag'][:, 1:4], g['ref_mag'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['gps'][:, 1:7], g['ref_gps'], rtol=1e-15, atol=1e-13)


def test_hot_kernels_keep_two_wavefronts_per_simd():
    """The launch geometry counts on two resident wavefronts per SIMD for every hot kernel (csrc/mc_kernel.hip, launch3).
    hipcc reports registers / scratch / occupancy per kernel at build time (build/<file>.resources.txt): no hot
    kernel may drop to one wavefront, park registers in AGPRs or spill more than a few words.  (A variant that grew
    to 256 VGPRs + 4 AGPRs once ran 30 % slower at 262 144 runs without any functional symptom.)"""
    build = os.path.join(PKG, 'build')
    kernels = {}
    for fn in ('mc_kernel', 'mc_kernel_f32', 'allan', 'stats'):
        path = os.path.join(build, fn + '.resources.txt')
        assert os.path.exists(path), 'run gnss-ins-sim_amd/build.py (it writes %s)' % path
        cur = None
        for line in open(path):
            k, _, v = line.strip().partition(':')
            if k == 'Function Name':
                cur = kernels.setdefault(v.strip(), {})
            elif cur is not None and v.strip():
                cur[k.split('[')[0].strip()] = v.strip()
    checked = 0
    split_seen = {}
    for name, r in kernels.items():         # Itanium-mangled: ...9mc_kernelILi<rf>ELi<algos>ELb<given>ELb<general>EEEv...
        occ, agpr, scratch = int(r['Occupancy']), int(r['AGPRs']), int(r['ScratchSize'])
        hot = any(t in name for t in ('9mc_kernelI', '15mc_kernel_splitI', '13mc_kernel_f32I', '19mc_kernel_f32_splitI',
                                      'allan_level_kernel', 'process_stats_kernel', '13series_kernelI'))
        if not hot:
            continue
        checked += 1
        two_algos = 'ILi0ELi3E' in name or 'ILi1ELi3E' in name
        assert occ >= 2, '%s: %d wavefront(s) per SIMD' % (name, occ)
        if '15mc_kernel_splitI' in name or '19mc_kernel_f32_splitI' in name:
            # <RF, ALGOS, WD, PROD, KEEP[, VIB]>: 256 consumer + PROD x 256 producer threads, 1 + PROD wavefronts per SIMD
            m = re.search(r'ELi(\d)ELb[01]E(?:Lb[01]E)?EEv', name)
            assert m, name
            prod = int(m.group(1))
            split_seen[prod] = split_seen.get(prod, 0) + 1
            assert occ >= 1 + prod, '%s: %d wavefront(s) per SIMD' % (name, occ)
            if prod == 3:
                assert scratch == 0, '%s: %d bytes of scratch' % (name, scratch)
        assert agpr == 0, '%s parks %d registers in AGPRs' % (name, agpr)
        # the vibration variants (<RF, ALGOS, false, true, PS, true>, Sim(env=...)) are not hot kernels: they may spill
        vib = re.search(r'9mc_kernelILi\dELi\dELb0ELb1ELi\dELb1EEE', name) is not None
        assert scratch <= (192 if vib else 128 if two_algos else 32), '%s: %d bytes of scratch per lane' % (name, scratch)
    assert checked >= 40, checked
    assert split_seen.get(2, 0) >= 8 and split_seen.get(3, 0) >= 4, split_seen


def test_native_pathgen_random_profiles_vs_reference():
    """24 random motion definitions (2-7 segments of command types 1-5, random mobility limits, 20-125 Hz, GPS at 1-10 Hz with
    random visibility, both frames) executed by the unmodified reference (tests/golden/make_golden.py
    truth_random_profiles): the native generator gives the same sample counts -- the segment-completion tests are threshold
    compares, a different rounding anywhere would change n -- and the same rows."""
    import ginsim
    g = load_golden('truth_random_profiles')
    worst = 0.0
    for i in range(int(g['count'])):
        c = {k[4:]: g[k] for k in g if k.startswith('c%02d_' % i)}
        r = ginsim.pathgen(c['ini'], c['md'], float(c['fs']), float(c['fs_gps']), c['mob'], int(c['rf']), gps=True)
        assert r['imu'].shape[0] == int(c['n']) and r['gps'].shape[0] == int(c['m']), (i, r['imu'].shape, int(c['n']), r['gps'].shape, int(c['m']))
        k, kg = c['k'], c['kg']
        # measured: nav / gps / odo identical to the last bit, imu within 3.6e-15 (one ulp of its terms)
        np.testing.assert_allclose(r['imu'][k], c['imu'], rtol=0, atol=2e-14, err_msg='profile %d imu' % i)
        assert np.array_equal(r['nav'][k], c['nav']), 'profile %d nav' % i
        if int(c['m']):
            assert np.array_equal(r['gps'][kg], c['gps']), 'profile %d gps' % i
        assert np.array_equal(r['odo'][k], c['odo']), 'profile %d odo' % i
        worst = max(worst, float(np.abs(r['imu'][k] - c['imu']).max()))
    assert worst < 2e-14


@pytest.mark.skipif(not os.path.isfile('/root/reference/gnss_ins_sim/geoparams/geomag.py'), reason='needs the reference checkout (build container)')
def test_nine_axis_hook_uses_the_references_own_wmm(monkeypatch):
    """VERDICT r02 missing 6: a 9-axis Sim without geo_mag_n evaluates the World Magnetic Model once on the host exactly as
    pathgen.py:164-168 does -- with the reference's own geomag.py when a checkout is reachable -- and says what to do otherwise."""
    import datetime
    import importlib.util
    from gnss_ins_sim.geoparams import geoparams
    monkeypatch.delenv('GNSS_INS_SIM_REFERENCE', raising=False)
    lat, lon, alt, when = 0.5584, 2.0944, 12.0, datetime.date(2025, 3, 1)
    clean = [p for p in sys.path if not os.path.isfile(os.path.join(p, 'gnss_ins_sim', 'geoparams', 'geomag.py'))]
    monkeypatch.setattr(sys, 'path', clean)
    assert geoparams.reference_geomag_n(lat, lon, alt, when) is None              # no checkout reachable
    monkeypatch.setenv('GNSS_INS_SIM_REFERENCE', '/root/reference')
    got = geoparams.reference_geomag_n(lat, lon, alt, when)
    spec = importlib.util.spec_from_file_location('_ref_geomag_direct', '/root/reference/gnss_ins_sim/geoparams/geomag.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import math
    r = mod.GeoMag('WMM.COF').GeoMag(lat / (math.pi / 180), lon / (math.pi / 180), alt, when)
    np.testing.assert_array_equal(got, np.array([r.bx, r.by, r.bz]) / 1000.0)
    assert 20.0 < np.linalg.norm(got) < 70.0                                       # uT


def test_summary_vector_text_equals_numpy_str():
    """The summary writes str(statistic vector) tens of thousands of times for a large batch; sim_data.vec_str reproduces
    numpy's array printer (positional / exponent form, common widths) without its per-call set-up.  Same text on 40 000 vectors
    of every magnitude, with zeros, negative zeros, rounded values, the thresholds of the exponent form; anything it does not
    cover goes to str()."""
    from gnss_ins_sim.sim import sim_data
    rng = np.random.default_rng(7)
    cases = []
    for scale in (1e-12, 1e-7, 1e-5, 1e-4, 1e-3, 1e-2, 1, 10, 1e3, 1e5, 1e7, 1e8, 1e9, 1e12):
        for _ in range(2800):
            a = rng.normal(size=3) * scale * 10 ** rng.uniform(-2, 2, size=3)
            if rng.random() < 0.2:
                a[rng.integers(3)] = 0.0
            if rng.random() < 0.2:
                a = np.round(a, int(rng.integers(0, 6)))
            if rng.random() < 0.1:
                a = np.abs(a)
            cases.append(a)
    cases += [np.zeros(3), np.array([1., 2., 3.]), np.array([0.5, -0.25, 100.]), np.array([1e-4, 1e-4, 1e-4]),
              np.array([9.99999999e7, 1, 1]), np.array([1e8, 1, 1]), np.array([-0.0, 1.0, 2.0]), np.array([1e-5, 0, 0]),
              np.array([123456789.123, 0, 0]), np.array([0.1, 100.0, 0.1]), np.array([0.1, 100.1, 0.1]),
              np.array([1.0, 1000.0, 1.0]), np.array([1.0, 1000.1, 1.0]), np.array([2.5]), np.array([1.0, -2.0]),
              np.array([np.nan, 1.0, 2.0]), np.array([np.inf, 1.0, -2.0]), np.arange(6.0), np.array([1, 2, 3]),
              # three-digit exponents next to two-digit ones, zeros of both signs in the exponent form, one element
              np.array([1e-300, 1.0, 2.0]), np.array([1e100, 1e-100, 3.0]), np.array([1.5e-5, 0.0, -0.0]), np.array([-2.5e-7]),
              np.array([1.23456789e-5, -1e-5, 1e-5]), np.array([5e-324, 1.0, 1.0]), np.array([1.7976931348623157e308, 1.0, -1.0])]
    # vectors of 4-8 elements can exceed numpy's line width and are wrapped by str(); the fast writer does not
    # wrap, so they must take the str() path -- long values of every size that would not fit one line
    for size in range(4, 9):
        for scale in (1e-7, 1.0, 1e9):
            for _ in range(60):
                cases.append(rng.normal(size=size) * scale * 10 ** rng.uniform(-2, 2, size=size))
    assert sim_data.default_print_options()
    for a in cases:
        assert sim_data.vec_str(a) == str(a), repr(a)
    assert any('\n' in str(a) for a in cases)             # the wrapped form really occurs among them
    with np.printoptions(precision=3):
        assert not sim_data.default_print_options()
        a = np.array([1.23456789, 2.0, 3.0])
        assert sim_data.vec_str(a, sim_data.default_print_options()) == str(a)


def test_run_stats_first_keys_are_the_first_runs_of_every_algorithm():
    """What a truncated summary lists: keys in (algorithm, run number) order, made without building the rest; the mapping
    itself still answers every key."""
    from gnss_ins_sim.sim import sim_data
    a = np.arange(30.0).reshape(10, 3)
    rs = sim_data.RunStats([('odo', 0, a[:4]), ('free', 0, a[:6]), ('free', 6, a[6:])])
    assert len(rs) == 14
    assert rs.first_keys(8) == ['free_0', 'free_1', 'free_2', 'free_3', 'free_4', 'free_5', 'free_6', 'free_7']
    assert rs.first_keys(12) == ['free_%d' % i for i in range(10)] + ['odo_0', 'odo_1']
    assert rs.first_keys(100) == ['free_%d' % i for i in range(10)] + ['odo_%d' % i for i in range(4)]
    assert np.array_equal(rs['free_7'], a[7]) and np.array_equal(rs['odo_3'], a[3])
    assert sorted(rs.keys()) == sorted(rs.first_keys(100))


@pytest.mark.parametrize('rf', [0, 1])
def test_path_gen_under_its_reference_name(rf):
    """`from gnss_ins_sim.pathgen import pathgen; pathgen.path_gen(ini, motion_def, output_def, mobility, ref_frame, magnet)`
    (pathgen.py:26-329) with the drop-in package ahead on the path: the reference's signature, its result dict ('status', 'imu',
    'nav', 'mag', 'gps', 'odo'; [] for what is switched off) and its exceptions, over ginsim_pathgen -- against the truth the
    executed reference produced (t2 goldens)."""
    from gnss_ins_sim.pathgen import pathgen
    g = load_golden('t2_turn_rf%d' % rf)
    k = g['rows']
    md0 = g['motion_def'].copy()
    out_def = np.array([[1.0, 100.0], [1.0, 10.0], [1.0, 100.0]])
    r = pathgen.path_gen(g['ini_pva'], md0, out_def, g['mobility'], rf, False)
    assert sorted(r) == ['gps', 'imu', 'mag', 'nav', 'odo', 'status'] and r['status'] is True and r['mag'] == []
    assert np.array_equal(md0, g['motion_def']) and out_def[1, 1] == 10.0          # the arguments are left alone
    assert r['imu'].shape == (int(g['n']), 7) and r['nav'].shape == (int(g['n']), 10) and r['odo'].shape == (int(g['n']), 5)
    np.testing.assert_allclose(r['imu'][:, 1:4], g['full_ref_accel'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['imu'][:, 4:7], g['full_ref_gyro'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(r['nav'][k, 1:4], g['ref_pos'], rtol=1e-15, atol=0)
    np.testing.assert_allclose(r['nav'][k, 4:7], g['ref_vel'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['nav'][k, 7:10], g['ref_att'], rtol=0, atol=1e-14)
    np.testing.assert_allclose(r['gps'][:, 1:7], g['ref_gps'], rtol=1e-15, atol=1e-13)
    np.testing.assert_allclose(r['gps'][:, 7], g['gps_vis'])
    np.testing.assert_allclose(r['odo'][k, 2], g['ref_odo'], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(r['imu'][:, 0], np.arange(int(g['n'])))           # index column = simulation count (osr 1)
    # GPS and odometer switched off: empty lists, as pathgen.py:99-104 initialises them
    off = pathgen.path_gen(g['ini_pva'], md0, np.array([[1.0, 100.0], [0.0, 10.0], [-1.0, 100.0]]), g['mobility'], rf)
    assert off['gps'] == [] and off['odo'] == [] and off['imu'].shape == r['imu'].shape
    # magnetometer truth with a supplied field vector (the WMM evaluation itself is an input)
    m9 = load_golden('t3_mag9_gps_rf%d' % rf)
    mag = pathgen.path_gen(g['ini_pva'], md0, out_def, g['mobility'], rf, True, geo_mag_n=m9['geo_mag_n'])
    np.testing.assert_allclose(mag['mag'][:, 1:4], m9['ref_mag'], rtol=0, atol=1e-13)
    # the reference's own errors (pathgen.py:118-126, 147)
    bad = md0.copy()
    bad[1, 7] = -1.0
    with pytest.raises(ValueError, match='negative time duration'):
        pathgen.path_gen(g['ini_pva'], bad, out_def, g['mobility'], rf)
    zero = md0.copy()
    zero[:, 7] = 0.0
    with pytest.raises(ValueError, match='must be above 0'):
        pathgen.path_gen(g['ini_pva'], zero, out_def, g['mobility'], rf)
    with pytest.raises(ValueError, match='3x2'):
        pathgen.path_gen(g['ini_pva'], md0, np.array([[1.0, 100.0], [1.0, 10.0]]), g['mobility'], rf)
    with pytest.raises(NotImplementedError):
        pathgen.path_gen(g['ini_pva'], md0, np.array([[2.0, 100.0], [1.0, 10.0], [1.0, 100.0]]), g['mobility'], rf)


def test_dispatch_queries_and_sensor_layout_rules_without_a_gpu():
    """ginsim_mc_variant / ginsim_mc_kernel_name are host code (they run the library's own dispatch without launching): which
    kernel serves a parameter block, and the ABI-4 rule that the series-major sensor layout is written by the time-parallel
    series kernels only."""
    import ctypes as C
    import ginsim
    from ginsim import _lib

    def params(**kw):
        p = _lib.McParams()
        p.n, p.runs, p.fs, p.ref_frame, p.algo_mask, p.n_ini = 1000, 65536, 100.0, 1, 1, 1
        for k in ('ini', 'ref_accel', 'ref_gyro', 'out_accel', 'out_gyro'):
            setattr(p, k, 4096)                       # never dereferenced: nothing is launched
        p.out_traj[0] = 4096
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def query(p):
        v = C.c_int32(-1)
        _lib.check(_lib.lib.ginsim_mc_variant(C.byref(p), C.byref(v)))
        buf = C.create_string_buffer(256)
        _lib.check(_lib.lib.ginsim_mc_kernel_name(C.byref(p), buf, 256))
        return v.value, buf.value.decode()

    assert query(params()) == (1, 'ginsim::mc_kernel_split<1, 1, false, 2, true, false>')                    # C2: the wave-specialised kernel
    assert query(params(given_sensors=1, in_gyro=4096, in_accel=4096)) == (0, 'ginsim::mc_kernel<1, 1, true, false, 0, false>')
    assert query(params(precision=1))[1].startswith('ginsim::f32::mc_kernel_f32_split<1, 1, false, 3,')
    # sensors only, few runs, long series: the time-parallel kernels -- with the series-major layout, or with one run (same thing)
    few = dict(algo_mask=0, runs=32, n=1440000)
    assert query(params(sensor_layout=1, **few)) == (2, 'ginsim::series_kernel<3>')                    # pass B of the simple sensor model
    assert query(params(sensor_layout=0, **dict(few, runs=1))) == (2, 'ginsim::series_kernel<3>')
    general = params(sensor_layout=1, **few)
    general.gyro.bias[1] = 1e-4                                                                        # a constant bias: the general model
    assert query(general) == (2, 'ginsim::series_kernel<1>')
    assert query(params(sensor_layout=0, **few))[0] == 0                                               # run-fastest layout: one lane per run
    for bad in (dict(sensor_layout=1), dict(sensor_layout=1, algo_mask=0, runs=2000, n=1440000), dict(sensor_layout=1, algo_mask=0, runs=32, n=1000),
                dict(sensor_layout=2, **few)):
        with pytest.raises(ValueError, match='sensor_layout'):
            query(params(**bad))
    # ABI 5: a vibration term (Sim(env=...)) is served by the vibration variants of the plain general-model kernels, whatever
    # the batch would otherwise run on (sensors only for few runs: pass B of the series kernels; fp32: the plain float kernel); given sensors refuse it
    v = ginsim.vibration({'type': 'random', 'x': 0.1, 'y': 0.1, 'z': 0.1}, 100.0, False)
    s = ginsim.vibration({'type': 'sinusoidal', 'x': 0.1, 'y': 0.1, 'z': 0.1, 'freq': 2.0}, 100.0, True)
    assert (s.type, s.random_phase) == (2, 1) and s.omega_dt == 2.0 * np.pi * 2.0 * (1.0 / 100.0) and v.type == 1
    assert query(params(vib_accel=v)) == (1, 'ginsim::mc_kernel_split<1, 1, true, 1, true, true>')        # C2's shape: one wavefront per SIMD otherwise
    assert query(params(vib_accel=v, runs=65537)) == (0, 'ginsim::mc_kernel<1, 1, false, true, 0, true>')     # larger batches: the plain kernel
    assert query(params(vib_gyro=s, ref_frame=0, algo_mask=3, ref_odo=4096)) == (0, 'ginsim::mc_kernel<0, 3, false, true, 0, true>')
    assert query(params(vib_gyro=s, ref_frame=0, ref_nav=4096, out_proc=(C.c_void_p * 2)(4096, None), proc_pos_ned=1)) == \
        (0, 'ginsim::mc_kernel<0, 1, false, true, 2, true>')
    assert query(params(vib_accel=v, sensor_layout=1, **few)) == (2, 'ginsim::series_kernel<2>')           # per-sample term: time-parallel too
    assert query(params(vib_accel=v, **few)) == (0, 'ginsim::mc_kernel<1, 0, false, true, 0, true>')
    assert query(params(vib_accel=v, precision=1)) == (0, 'ginsim::f32::mc_kernel_f32<1, 1, false, true, true>')
    with pytest.raises(ValueError, match='vibration'):
        query(params(vib_accel=v, given_sensors=1, in_gyro=4096, in_accel=4096))


def test_env_strings_parse_like_the_reference_and_bad_ones_raise():
    """Sim.__parse_env (ins_sim.py:642-701): units ('g' = 9.8 m/s^2, 'd' = deg/s), the frequency of a sinusoidal model, the
    exceptions of malformed strings; a PSD array is parsed (cut at fs/2)."""
    from gnss_ins_sim.sim import imu_model
    from gnss_ins_sim.sim import ins_sim
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='low-accuracy', axis=6, gps=False)
    sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, env=None, algorithm=None)
    for name in ('t3_vib_random_rf1', 't3_vib_sin_rf0', 't3_vib_mixed_rf1'):
        g = load_golden(name)
        for sensor in ('acc', 'gyro'):
            v = sim._parse_env(str(g['env_' + sensor]))
            assert v['type'] == str(g['vib_%s_type' % sensor])
            assert np.array_equal([v['x'], v['y'], v['z']], g['vib_%s_amp' % sensor])       # the same products, bit for bit
            assert v.get('freq', 0.0) == float(g['vib_%s_freq' % sensor])
    assert sim._parse_env(None) is None
    assert sim._parse_env('[1 2 3]G-RANDOM') == {'type': 'random', 'x': 9.8, 'y': 19.6, 'z': 9.8 * 3.0}
    for bad, exc in (('[1 2 3]g-sinusoidal', ValueError), ('[1 2 3]-fastHz-sinusoidal', ValueError), ('[1 2 3]g-square', ValueError),
                     ('[1 2]-random', ValueError), ('[a b c]-random', ValueError), (5, TypeError), (np.zeros((4, 2)), TypeError)):
        with pytest.raises(exc):
            sim._parse_env(bad)
    psd = np.array([[1.0, 1e-3, 1e-3, 1e-3], [20.0, 2e-3, 2e-3, 2e-3], [60.0, 1e-3, 1e-3, 1e-3]])
    v = sim._parse_env(psd)
    assert v['type'] == 'psd' and v['freq'].tolist() == [1.0, 20.0]                          # 60 Hz is above fs / 2
    # the PSD goldens: what the reference's own __parse_env made of the arrays (the cut at fs / 2), and the host half of
    # time_series_from_psd (ginsim.psd_amplitudes: period, interpolation, halving, a = sqrt(sxx N fs))
    import ginsim
    g = load_golden('t3_vib_psd_rf1')
    for sensor in ('acc', 'gyro'):
        v = sim._parse_env(g['env_' + sensor].copy())
        assert v['type'] == 'psd' and np.array_equal(v['freq'], g['vib_%s_freq' % sensor])
        assert np.array_equal([v['x'], v['y'], v['z']], g['vib_%s_amp' % sensor])
    acc = sim._parse_env(g['env_acc'].copy())
    N, amp, on_grid = ginsim.psd_amplitudes(acc, 100.0, 1000)
    assert (N, amp.shape, on_grid) == (1000, (3, 501), False) and acc['freq'][-1] == 40.0        # the 60 Hz row is gone
    sxx = np.interp(np.linspace(0, 50.0, 501), acc['freq'], acc['y'])
    sxx[1:500] *= 0.5
    assert np.array_equal(amp[1], np.sqrt(sxx * 1000 * 100.0)) and amp[1][-1] == np.sqrt(acc['y'][-1] * 1e5)   # beyond the last row np.interp holds it; the end bins are not halved
    gyr = sim._parse_env(g['env_gyro'].copy())
    before = gyr['x'].copy()
    N, amp, on_grid = ginsim.psd_amplitudes(gyr, 100.0, 1000)
    assert on_grid and np.array_equal(amp[0], np.sqrt(before * 1000 * 100.0)) and np.array_equal(gyr['x'], before)   # nothing halved, nothing touched
    assert ginsim.psd_amplitudes(gyr, 100.0, 333)[0] == 334 and ginsim.psd_amplitudes(gyr, 100.0, 17000)[0] == 16384
    assert ginsim.psd_amplitudes(gyr, 90.0, 1000) is None                                            # fs < 2 freq[-1]: the reference returns zeros
    with pytest.raises(ValueError, match='psd'):
        ginsim.vibration(gyr, 100.0, False)                                                          # a series per run, not a term: MonteCarloJob makes it
    s2 = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, env={'acc': psd}, algorithm=None, precision='f32')
    with pytest.raises(NotImplementedError, match='fp64'):
        s2.run(1)


def test_unconfigured_sim_spreads_large_batches_over_every_gpu(monkeypatch):
    """The default of Sim(devices=None): every visible GPU for a batch of >= 2^30 sample x run products -- unless something says
    this process owns ONE GPU (device=, $LOCAL_RANK, a process group) or $GINSIM_DEVICES decides.  The decision is host logic:
    checked here with a pretended device count (the reference's loop being sharded: ins_sim.py:490-506)."""
    import ginsim
    from ginsim import multi
    from gnss_ins_sim.sim import imu_model, ins_sim

    class FakeSet(object):
        def __init__(self, devices):
            self.devices = multi.parse_devices(devices)

    monkeypatch.setattr(ginsim, 'device_count', lambda: 8)
    monkeypatch.setattr(ginsim.engine, 'device_count', lambda: 8)
    monkeypatch.setattr(multi, 'device_count', lambda: 8)
    monkeypatch.setattr(multi, 'DeviceSet', FakeSet)
    monkeypatch.setattr(ginsim, 'default_context', lambda: 'ctx0')
    monkeypatch.delenv('LOCAL_RANK', raising=False)
    monkeypatch.delenv('GINSIM_DEVICES', raising=False)
    csv = os.path.join(PKG, 'motion_profiles', 'turn_90deg.csv')
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    mk = lambda **kw: ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, **kw)
    big, small = 2 ** 30, 65536 * 1000
    assert mk()._context(small) == 'ctx0'                                   # C2: one GPU is faster than setting up eight
    assert mk()._context(big).devices == list(range(8))                    # C3 / C4 sized: the whole node
    assert mk()._context(big, distributed=True) == 'ctx0'                  # one process per GPU: the launcher's split
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert mk()._context(big) == 'ctx0'
    monkeypatch.delenv('LOCAL_RANK')
    monkeypatch.setenv('GINSIM_DEVICES', 'one')
    assert mk()._context(big) == 'ctx0'
    monkeypatch.setenv('GINSIM_DEVICES', '2,5')
    assert mk()._context(small).devices == [2, 5]
    monkeypatch.setenv('GINSIM_DEVICES', 'all')
    assert mk()._context(small).devices == list(range(8))
    assert mk(devices=[1, 1])._context(small).devices == [1, 1]            # the argument wins over the environment
    with pytest.raises(ValueError, match='not both'):
        mk(devices=[0], device=0)._context(small)


def test_reported_kernel_names_are_the_compiled_kernels():
    """ginsim_mc_kernel_name is what the bench and the profiles use to find a launch's counters in rocprofv3's output, so the name it
    reports must be the (demangled) name of a kernel the library really contains -- a template parameter added to a kernel without its
    name string silently detaches the PMC traffic from the roofline (round 5: `traffic: null` in one bench run)."""
    import ctypes as C
    import shutil
    import ginsim
    from ginsim import _lib
    if not shutil.which('c++filt'):
        pytest.skip('no c++filt to demangle the build\'s kernel list')
    mangled = []
    for fn in ('mc_kernel', 'mc_kernel_f32'):
        for line in open(os.path.join(PKG, 'build', fn + '.resources.txt')):
            if line.startswith('Function Name:'):
                mangled.append(line.split(':', 1)[1].strip())
    out = subprocess.run(['c++filt'], input='\n'.join(mangled), stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
    compiled = {re.sub(r'\(.*$', '', l.replace('void ', '')).strip() for l in out.splitlines()}

    def name(**kw):
        p = _lib.McParams()
        p.n, p.runs, p.fs, p.ref_frame, p.algo_mask, p.n_ini = 1000, 65536, 100.0, 1, 1, 1
        for k in ('ini', 'ref_accel', 'ref_gyro', 'out_accel', 'out_gyro', 'ref_odo', 'ref_nav'):
            setattr(p, k, 4096)
        p.out_traj[0] = 4096
        for k, v in kw.items():
            setattr(p, k, v)
        buf = C.create_string_buffer(256)
        _lib.check(_lib.lib.ginsim_mc_kernel_name(C.byref(p), buf, 256))
        return buf.value.decode()

    v = ginsim.vibration({'type': 'random', 'x': 0.1, 'y': 0.1, 'z': 0.1}, 100.0, False)
    none = C.c_void_p * 2
    cases = [dict(), dict(ref_frame=0), dict(runs=262144), dict(precision=1), dict(precision=1, runs=262144), dict(algo_mask=3),
             dict(algo_mask=2), dict(given_sensors=1, in_gyro=4096, in_accel=4096), dict(vib_accel=v), dict(vib_accel=v, runs=262144),
             dict(vib_accel=v, ref_frame=0), dict(out_accel=None, out_gyro=None, out_traj=none(None, None)),
             dict(ref_frame=0, out_accel=None, out_gyro=None, out_traj=none(None, None)),
             dict(ref_frame=0, runs=262144, out_accel=None, out_gyro=None, out_traj=none(None, None), out_proc=none(4096, None)),
             dict(ref_frame=0, runs=262144, out_accel=None, out_gyro=None, out_traj=none(None, None), out_proc=none(4096, None), proc_pos_ned=1),
             dict(out_accel=None, out_gyro=None, out_traj=none(None, None), out_proc=none(4096, None)),
             dict(precision=1, out_accel=None, out_gyro=None, out_traj=none(None, None))]
    # the plain-sum forms of the statistics kernels whose shift is per launch (proc_plain_sums = 1: ref_frame 0 free integration only)
    plain = [dict(c, proc_plain_sums=1) for c in cases if c.get('out_proc') is not None]
    assert name(**plain[0]) == 'ginsim::mc_kernel<0, 1, false, false, 3, false>' and name(**cases[13]) == 'ginsim::mc_kernel<0, 1, false, false, 1, false>'
    assert name(**plain[1]) == 'ginsim::mc_kernel<0, 1, false, true, 4, false>'
    assert name(**plain[2]) == name(**cases[15])                # ref_frame 1: the shift is per lane, there is no plain form
    cases += plain
    for kw in cases:
        k = name(**kw)
        assert k in compiled, 'ginsim_mc_kernel_name reports %r for %r, which is not a compiled kernel' % (k, kw)


def test_dropin_launcher_puts_the_drop_in_in_front_of_a_checkout_s_own_packages(tmp_path):
    """Round 6 (found by executing the reference's demo_free_integration.py byte for byte): `python script.py` makes the SCRIPT's
    directory sys.path[0], and in a checkout of the reference that directory holds the reference's own gnss_ins_sim/ and
    demo_algorithms/ -- $PYTHONPATH alone does not shadow them.  gnss-ins-sim_amd/dropin.py runs the script with the drop-in first
    and names the checkout for the fall-through.  Here: a stand-in "checkout" whose packages would raise on import."""
    import subprocess
    import sys
    co = tmp_path / 'checkout'
    for pkg in ('gnss_ins_sim', 'gnss_ins_sim/sim', 'demo_algorithms'):
        (co / pkg).mkdir(parents=True)
        (co / pkg / '__init__.py').write_text('raise ImportError("the checkout\'s own package was imported")\n')
    (co / 'gnss_ins_sim' / 'sim' / 'ins_sim.py').write_text('raise ImportError("the checkout\'s own ins_sim was imported")\n')
    (co / 'demo.py').write_text(
        'import os, sys\n'
        'from gnss_ins_sim.sim import imu_model\n'
        'from demo_algorithms import free_integration\n'
        'print("IMU", imu_model.__file__)\n'
        'print("ALGO", free_integration.__file__)\n'
        'print("REF", os.environ.get("GNSS_INS_SIM_REFERENCE"))\n'
        'print("ARGV", sys.argv)\n')
    env = {k: v for k, v in os.environ.items() if k not in ('PYTHONPATH', 'GNSS_INS_SIM_REFERENCE')}
    out = subprocess.run([sys.executable, os.path.join(PKG, 'dropin.py'), 'demo.py', '--flag'], cwd=str(co), env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, universal_newlines=True, timeout=120)
    assert out.returncode == 0, out.stdout
    lines = dict(l.split(' ', 1) for l in out.stdout.splitlines() if ' ' in l)
    assert os.path.abspath(lines['IMU']).startswith(os.path.abspath(PKG)) and os.path.abspath(lines['ALGO']).startswith(os.path.abspath(PKG))
    assert lines['REF'] == str(co) and lines['ARGV'] == "['demo.py', '--flag']"
    # the plain way fails exactly as a user in a checkout would see it: the checkout's packages win
    plain = subprocess.run([sys.executable, 'demo.py'], cwd=str(co), env=dict(env, PYTHONPATH=PKG), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, universal_newlines=True, timeout=120)
    assert plain.returncode != 0 and "the checkout's own package was imported" in plain.stdout


def test_placed_arena_logic_on_the_host(tmp_path):
    """The device-free parts of the placed arena (csrc/placed_logic.hpp: the first-fit free list with owners, the plan of how many
    chunks each class gives, the order of the classes over the stripes) compiled with g++ and checked on the CPU
    (tests/cpp/placed_logic_check.cpp): 20 000 random carve / give-back operations without overlap or lost bytes, the three-quarters
    rule, planes of exactly three stripes spread over the classes."""
    import subprocess
    exe = str(tmp_path / 'placed_logic_check')
    subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-Wall', '-Werror'] + SANITIZE + ['-I' + os.path.join(PKG, 'csrc'), '-o', exe,
                    os.path.join(REPO, 'tests', 'cpp', 'placed_logic_check.cpp')], check=True, timeout=300)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', out.stdout


# GPU AddressSanitizer is not available on the pool: the sanitizers run on the CPU builds of the host-side code
SANITIZE = ['-fsanitize=address,undefined', '-fno-sanitize-recover=all']


def test_pathgen_under_the_sanitizers(tmp_path):
    """csrc/pathgen.cpp (the truth generator, host code of libginsim.so: pathgen.py:26-329) compiled with g++ and the address and
    undefined-behaviour sanitizers, driven by tests/cpp/pathgen_sanitize_check.cpp: every command type, both frames, GPS and
    magnetometer rows on and off, three rates, into heap blocks of exactly the sizes include/ginsim.h:130-137 asks for; the
    refusals; the ABI-6 leaves.  (The numbers are held to the reference by test_native_pathgen_*.)"""
    import subprocess
    exe = str(tmp_path / 'pathgen_sanitize_check')
    subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-ffp-contract=off', '-Wall'] + SANITIZE + ['-I' + os.path.join(REPO, 'include'), '-o', exe,
                    os.path.join(PKG, 'csrc', 'pathgen.cpp'), os.path.join(REPO, 'tests', 'cpp', 'pathgen_sanitize_check.cpp')],
                   check=True, timeout=300)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', out.stdout
