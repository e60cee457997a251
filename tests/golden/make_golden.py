#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING the unmodified reference.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

What is produced (all from reference code, nothing hand-typed):
  t1_fixture_{bosch,nxp}.npz   FreeIntegration.run on the reference's logged-IMU fixtures
                               (recipe of demo_free_integration_openimu.py:31-53).
  t2_turn_rf{0,1}.npz          noise-free closed loop on the 90-degree turn: path_gen truth +
                               FreeIntegration / odo-FreeIntegration outputs (decimated).
  t2_long_drive_rf0.npz        path_gen truth of long_drive @200 Hz with GPS@10 Hz + odo, and the
                               noise-free FreeIntegration end state (decimated).
  t3_*.npz                     injected-noise end-to-end: Sim.run(R) with np.random.randn replaced by
                               oracle.ref_shim (the engine's Philox normals in the reference's call
                               order) -> per-run sensors, algorithm outputs, end-point statistics.
  allan_ref.npz                allan.allan_var on a fixed Philox-generated series.
Motion profiles (numeric workload definitions) are re-emitted as
gnss-ins-sim_amd/motion_profiles/*.csv so bench.py and the GPU tests can run without the reference.
"""
import os
import sys
import math

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

from gnss_ins_sim.sim import imu_model, ins_sim            # noqa: E402  (reference)
from gnss_ins_sim.pathgen import pathgen                    # noqa: E402
from gnss_ins_sim.allan import allan                        # noqa: E402
from demo_algorithms import free_integration, free_integration_odo   # noqa: E402
from oracle import philox                                   # noqa: E402
from oracle.ref_shim import RandnShim, injected            # noqa: E402

D2R = math.pi / 180
MOTION = REF + '/demo_motion_def_files/'
SEED = 20260923


OUT = os.environ.get('GINSIM_GOLDEN_OUT')        # --check: write into a scratch directory instead of the tree


def save(name, **arrays):
    path = os.path.join(OUT or HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def rows(n, stride):
    idx = list(range(0, n, stride))
    for k in (n - 2, n - 1):
        if k not in idx and k >= 0:
            idx.append(k)
    return np.array(sorted(idx))


def read_ini(csv):
    ini = np.genfromtxt(csv, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= D2R
    ini[6:9] *= D2R
    return ini


def parsed_motion(csv):
    """ini_pva / motion_def exactly as Sim.__parse_motion produces them (ins_sim.py:578-610)."""
    s = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=None)
    return s._Sim__parse_motion()


def emit_profile(src, dst, comment):
    ini = np.genfromtxt(src, delimiter=',', skip_header=1, max_rows=1)
    seg = np.genfromtxt(src, delimiter=',', skip_header=3)
    if seg.ndim == 1:
        seg = seg.reshape(1, -1)
    seg[np.isnan(seg)] = 0.0
    if OUT:
        dst = os.path.join(OUT, 'motion_profiles', os.path.basename(dst))
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, 'w') as f:
        f.write('lat_deg,lon_deg,alt_m,vbx_mps,vby_mps,vbz_mps,yaw_deg,pitch_deg,roll_deg  # %s\n' % comment)
        f.write(','.join(repr(float(v)) for v in ini) + '\n')
        f.write('type,yaw,pitch,roll,vbx,vby,vbz,duration_s,gps_visible\n')
        for r in seg:
            f.write(','.join(repr(float(v)) for v in r[:9]) + '\n')


# ------------------------------------------------------------------ T1: logged-data fixtures
def t1_run(name, ini, gyro, accel, stride):
    k = rows(gyro.shape[0], stride)
    out = {}
    for tag, use_g, erot in (('extg', True, False), ('wgs', False, True)):
        algo = free_integration.FreeIntegration(ini if use_g else ini[0:9], earth_rot=erot)
        algo.run([0, 100.0, gyro.copy(), accel.copy()])
        att, pos, vel = algo.get_results()
        out.update({'att_' + tag: att[k], 'pos_' + tag: pos[k], 'vel_' + tag: vel[k]})
    algo = free_integration.FreeIntegration(ini[0:9])
    algo.run([1, 100.0, gyro.copy(), accel.copy()])
    att, pos, vel = algo.get_results()
    out.update({'att_rf1': att[k], 'pos_rf1': pos[k], 'vel_rf1': vel[k]})
    save('t1_fixture_' + name, rows=k, ini=ini, gyro=gyro, accel=accel, fs=100.0, **out)
    return att


def t1_fixture(name):
    d = REF + '/demo_data_files/%s/' % name
    ini = np.genfromtxt(d + 'ini.txt', delimiter=',')
    ini[0:2] *= D2R
    ini[6:9] *= D2R
    gyro = np.genfromtxt(d + 'gyro-0.csv', delimiter=',', skip_header=1) * D2R   # file is deg/s
    accel = np.genfromtxt(d + 'accel-0.csv', delimiter=',', skip_header=1)
    t1_run(name, ini, gyro, accel, 10)


def t1_rates():
    """Both plugins as the host calls them (FreeIntegration.run(set_of_input), free_integration.py:63-174 and
    free_integration_odo.py:63-160) on band-limited random sensor records at 50, 200 and 400 Hz (the other given-data
    fixtures are all 100 Hz), both frames, with and without Earth rotation / external gravity."""
    rng = np.random.RandomState(77)
    out = {}
    for ci, fs in enumerate((50.0, 200.0, 400.0)):
        n = 1600
        t = np.arange(n) / fs
        def band(scale, m=4):
            f = rng.uniform(0.05, 2.0, size=(m, 3))
            ph = rng.uniform(0, 2 * np.pi, size=(m, 3))
            amp = rng.uniform(0.2, 1.0, size=(m, 3)) * scale
            return sum(amp[k] * np.sin(2 * np.pi * f[k] * t[:, None] + ph[k]) for k in range(m))
        gyro = band(8.0 * D2R)
        accel = band(0.4) + np.array([0.0, 0.0, -9.79])
        odo = 6.0 + band(1.5)[:, 0]
        ini = np.array([rng.uniform(-50, 50) * D2R, rng.uniform(-170, 170) * D2R, rng.uniform(0, 300), rng.uniform(2, 12), 0.0, 0.0,
                        rng.uniform(-180, 180) * D2R, rng.uniform(-4, 4) * D2R, rng.uniform(-4, 4) * D2R, 9.79 + rng.uniform(-0.01, 0.01)])
        k = rows(n, 16)
        key = 'c%d_' % ci
        out.update({key + 'fs': fs, key + 'ini': ini, key + 'gyro': gyro, key + 'accel': accel, key + 'odo': odo, key + 'rows': k})
        for tag, rf, use_g, erot in (('extg', 0, True, False), ('wgs', 0, False, True), ('rf1', 1, False, True)):
            for plug, mod, series in (('free', free_integration, accel), ('odo', free_integration_odo, odo)):
                algo = mod.FreeIntegration(ini.copy() if use_g else ini[0:9].copy(), earth_rot=erot)
                algo.run([rf, fs, gyro.copy(), series.copy()])
                att, pos, vel = algo.get_results()
                out.update({key + '%s_%s_att' % (plug, tag): att[k], key + '%s_%s_pos' % (plug, tag): pos[k], key + '%s_%s_vel' % (plug, tag): vel[k]})
    save('t1_rates', count=3, **out)


def t1_tumble():
    """Synthetic body rates that drive the pitch through +-90 deg (the fold of attitude.euler_update_zyx,
    attitude.py:700-712) and yaw / roll through +-180 deg (the single 2 pi wrap, :713-720) several times.
    Every sample is kept: the branches are what is being pinned."""
    n = 900
    t = np.arange(n) / 100.0
    ini = np.array([32.0 * D2R, 120.0 * D2R, 0.0, 5.0, 0.0, 0.0, 170.0 * D2R, 78.0 * D2R, 0.0, 9.794])
    # a pure pitch rotation first (Euler kinematics only cross the pole when q = wz cos(roll) + wy sin(roll) ~ 0:
    # 78 deg + 0.61 deg per step overshoots 90 deg at step 20, -90 deg at step 315), then yaw / roll rates as well
    late = (t >= 4.0).astype(np.float64)
    gyro = np.stack([late * 40.0 * D2R * np.cos(2 * np.pi * t / 4.1), 61.0 * D2R * np.ones(n),
                     late * (70.0 * D2R + 10.0 * D2R * np.sin(2 * np.pi * t / 2.9))], axis=1)
    accel = np.stack([0.3 * np.sin(2 * np.pi * t / 3.0), 0.2 * np.cos(2 * np.pi * t / 1.7),
                      -9.794 + 0.1 * np.sin(2 * np.pi * t / 0.9)], axis=1)
    att = t1_run('tumble', ini, gyro, accel, 1)
    pit = att[:, 1]
    dy, dr = np.abs(np.diff(att[:, 0])), np.abs(np.diff(att[:, 2]))
    print('   tumble: min |cos(pitch)| = %.2e, %d folds, %d wraps, %d steps > 0.25 rad' % (
        np.min(np.abs(np.cos(pit))), int(np.sum((dy > 2) & (dy < 4) & (dr > 2) & (dr < 4))), int(np.sum(dy > 5) + np.sum(dr > 5)),
        int(np.sum(np.maximum(dy, dr) > 0.25))))


# ------------------------------------------------------------------ T2: noise-free closed loop
ZERO_IMU = {'gyro_b': np.zeros(3), 'gyro_arw': np.zeros(3), 'gyro_b_stability': np.zeros(3),
            'gyro_b_corr': np.array([100.0, 100.0, 100.0]),
            'accel_b': np.zeros(3), 'accel_vrw': np.zeros(3), 'accel_b_stability': np.zeros(3),
            'accel_b_corr': np.array([200.0, 200.0, 200.0])}


def t2_turn(ref_frame):
    csv = MOTION + 'motion_def-90deg_turn.csv'
    imu = imu_model.IMU(accuracy=dict(ZERO_IMU), axis=6, gps=True, odo=True,
                        odo_opt={'scale': 1.0, 'stdv': 0.0})
    ini = read_ini(csv)
    a_odo = free_integration_odo.FreeIntegration(ini.copy())
    a_fi = free_integration.FreeIntegration(ini.copy())
    sim = ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=ref_frame, imu=imu, algorithm=[a_odo, a_fi])
    sim.run(1)
    d = sim.dmgr
    n = d.time.data.shape[0]
    k = rows(n, 10)
    ini_pva, motion_def = parsed_motion(csv)
    save('t2_turn_rf%d' % ref_frame, fs=100.0, fs_gps=10.0, n=n, rows=k,
         ini_pva=ini_pva, motion_def=motion_def, mobility=ins_sim.high_mobility,
         ref_pos=d.ref_pos.data[k], ref_vel=d.ref_vel.data[k], ref_att=d.ref_att_euler.data[k],
         ref_accel=d.ref_accel.data[k], ref_gyro=d.ref_gyro.data[k], ref_odo=d.ref_odo.data[k],
         ref_gps=d.ref_gps.data, gps_time=d.gps_time.data, gps_vis=d.gps_visibility.data,
         full_ref_accel=d.ref_accel.data, full_ref_gyro=d.ref_gyro.data,
         odo_att=d.att_euler.data['algo0_0'][k], odo_pos=d.pos.data['algo0_0'][k],
         odo_vel=d.vel.data['algo0_0'][k],
         fi_att=d.att_euler.data['algo1_0'][k], fi_pos=d.pos.data['algo1_0'][k],
         fi_vel=d.vel.data['algo1_0'][k],
         ref_att_quat=d.ref_att_quat.data[k], fi_att_quat=d.att_quat.data['algo1_0'][k])


def t2_long_drive():
    csv = MOTION + 'motion_def-long_drive.csv'
    ini_pva, motion_def = parsed_motion(csv)
    fs, fs_gps = 200.0, 10.0
    output_def = np.array([[1.0, fs], [1.0, fs_gps], [1.0, fs]])
    r = pathgen.path_gen(ini_pva.copy(), motion_def.copy(), output_def, ins_sim.high_mobility,
                         ref_frame=0, magnet=False)
    n, m = r['imu'].shape[0], r['gps'].shape[0]
    k = rows(n, 997)
    kg = rows(m, 101)
    algo = free_integration.FreeIntegration(read_ini(csv))
    algo.run([0, fs, r['imu'][:, 4:7].copy(), r['imu'][:, 1:4].copy()])
    att, pos, vel = algo.get_results()
    save('t2_long_drive_rf0', fs=fs, fs_gps=fs_gps, n=n, m=m, rows=k, gps_rows=kg,
         ini_pva=ini_pva, motion_def=motion_def, mobility=ins_sim.high_mobility,
         imu=r['imu'][k], nav=r['nav'][k], gps=r['gps'][kg], odo=r['odo'][k],
         fi_att=att[k], fi_pos=pos[k], fi_vel=vel[k])


# ------------------------------------------------------------------ T3: injected noise, end to end
DRIVE = """ini lat (deg),ini lon (deg),ini alt (m),ini vx_body (m/s),ini vy_body (m/s),ini vz_body (m/s),ini yaw (deg),ini pitch (deg),ini roll (deg)
-33.9,151.2,55,4,0,0,-120,0,0
command type,yaw (deg),pitch (deg),roll (deg),vx_body (m/s),vy_body (m/s),vz_body (m/s),command duration (s),GPS visibility
1,0,0,0,0.8,0,0,5,1
5,45,0,0,12,0,0,8,1
1,0,0,0,0,0,0,3,0
3,-30,2,0,-3,0,0,6,1
"""

DEMO_IMU = {'gyro_b': np.array([0.0, 0.0, 0.0]),
            'gyro_arw': np.array([0.25, 0.25, 0.25]),
            'gyro_b_stability': np.array([3.5, 3.5, 3.5]),
            'gyro_b_corr': np.array([100.0, 100.0, 100.0]),
            'accel_b': np.array([0.0, 0.0, 0.0]),
            'accel_vrw': np.array([0.03119, 0.03009, 0.04779]),
            'accel_b_stability': np.array([4.29e-5, 5.72e-5, 8.02e-5]),
            'accel_b_corr': np.array([200.0, 200.0, 200.0])}


def psd_env(kind, unit=1.0):
    """(rows, 4) single-sided PSD arrays [freq, x, y, z] for Sim(env=...) (ins_sim.py:115-121, :686-697).
    'coarse': seven rows up to 60 Hz -- at fs = 100 Hz the rows above fs / 2 are cut by Sim.__parse_env and the rest is
    interpolated to the series' grid (time_series_from_psd.py:44-48); 'grid<L>': L rows on linspace(0, 50, L), the grid of a
    series of 2 (L - 1) samples at 100 Hz: no interpolation, the reference halves the array in place at every call (:49)."""
    if kind == 'coarse':
        f = np.array([0.0, 2.0, 5.0, 12.0, 25.0, 40.0, 60.0])
        x = np.array([0.0, 1e-4, 4e-4, 9e-4, 2e-4, 5e-5, 1e-5])
        return np.stack([f, unit * x, unit * 0.5 * x[::-1], unit * (x + 1e-4)], axis=1)
    L = int(kind[4:])
    f = np.linspace(0.0, 50.0, L)
    x = 2e-4 * np.exp(-((f - 14.0) / 6.0) ** 2) + 1e-5
    return np.stack([f, unit * x, unit * 0.3 * x, unit * (x[::-1] * 0.5)], axis=1)


PSD_ODD = """ini lat (deg),ini lon (deg),ini alt (m),ini vx_body (m/s),ini vy_body (m/s),ini vz_body (m/s),ini yaw (deg),ini pitch (deg),ini roll (deg)
31.5,120.4,10,3,0,0,20,0,0
command type,yaw (deg),pitch (deg),roll (deg),vx_body (m/s),vy_body (m/s),vz_body (m/s),command duration (s),GPS visibility
1,4,0,0,0.5,0,0,3.33,1
"""

PSD_LONG = """ini lat (deg),ini lon (deg),ini alt (m),ini vx_body (m/s),ini vy_body (m/s),ini vz_body (m/s),ini yaw (deg),ini pitch (deg),ini roll (deg)
31.5,120.4,10,6,0,0,-70,0,0
command type,yaw (deg),pitch (deg),roll (deg),vx_body (m/s),vy_body (m/s),vz_body (m/s),command duration (s),GPS visibility
1,0.5,0,0,0.02,0,0,120,1
1,-1,0,0,0,0,0,50,1
"""


def err_dict_arrays(prefix, e):
    return {prefix + k: np.array(v, dtype=np.float64) for k, v in e.items()}


def t3_case(name, ref_frame, accuracy, gps, odo_opt, algos, R, fs_gps=0.0, axis=6, csv=None, fs=100.0, env=None):
    """csv: a motion definition as text (default: the reference's 90-degree turn file, n = 1000 at 100 Hz).
    env: the reference's vibration dict ({'acc': '...', 'gyro': '...'}, ins_sim.py:108-124), string models only."""
    if csv is None:
        csv = MOTION + 'motion_def-90deg_turn.csv'
        ini = read_ini(csv)
        n, m = 1000, (100 if gps else 0)
    else:
        text = csv.strip().split('\n')
        ini = np.array([float(v) for v in text[1].split(',')])
        ini[0:2] *= D2R
        ini[6:9] *= D2R
        probe = ins_sim.Sim([fs, fs_gps, 0.0], csv, ref_frame=ref_frame, imu=None)
        ini_pva, motion_def = probe._Sim__parse_motion()
        r = pathgen.path_gen(ini_pva.copy(), motion_def.copy(), np.array([[1.0, fs], [1.0, fs_gps], [1.0, fs]]),
                             probe._Sim__parse_mode(None), ref_frame=ref_frame, magnet=False)
        n, m = r['imu'].shape[0], (r['gps'].shape[0] if gps else 0)
    imu = imu_model.IMU(accuracy=accuracy, axis=axis, gps=gps, odo=odo_opt is not None, odo_opt=odo_opt)
    objs = []
    for a in algos:
        mod = free_integration_odo if a == 'odo' else free_integration
        objs.append(mod.FreeIntegration(ini.copy()))
    env_given = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in (env or {}).items()}    # the reference halves a PSD given
    sim = ins_sim.Sim([fs, fs_gps, 0.0], csv, ref_frame=ref_frame, imu=imu, env=env, algorithm=objs)    # on the series' grid IN PLACE
    kind = lambda e: None if e is None else (('psd', min(n + n % 2, 16384) // 2 + 1) if isinstance(e, np.ndarray) else
                                             ('random' if 'random' in e.lower() else 'sinusoidal'))
    shim = RandnShim(SEED, n, imu.accel_err['b_corr'], imu.gyro_err['b_corr'], gps_m=m, mag=(axis == 9),
                     odo=odo_opt is not None, vib_acc=kind((env or {}).get('acc')), vib_gyro=kind((env or {}).get('gyro')))
    with injected(shim):
        sim.run(R)
    assert shim.run == R and not shim.queue
    d = sim.dmgr
    k = rows(n, 25)
    out = dict(seed=SEED, R=R, fs=fs, n=n, rows=k, ref_frame=ref_frame,
               ref_pos=d.ref_pos.data, ref_vel=d.ref_vel.data, ref_att=d.ref_att_euler.data,
               ref_accel=d.ref_accel.data, ref_gyro=d.ref_gyro.data, ini=ini)
    out.update(err_dict_arrays('accel_', imu.accel_err))
    out.update(err_dict_arrays('gyro_', imu.gyro_err))
    out['accel'] = np.stack([d.accel.data[r][k] for r in range(R)])
    out['gyro'] = np.stack([d.gyro.data[r][k] for r in range(R)])
    for sensor in ('acc', 'gyro'):          # the env strings and what the reference's own Sim.__parse_env makes of them
        if env and sensor in env:
            vd = sim._Sim__parse_env(env_given[sensor].copy() if isinstance(env_given[sensor], np.ndarray) else env_given[sensor])
            out['env_' + sensor] = np.array(env_given[sensor])              # as it was BEFORE the run
            out['vib_%s_type' % sensor] = np.array(vd['type'])
            out['vib_%s_amp' % sensor] = np.array([vd['x'], vd['y'], vd['z']], dtype=np.float64)
            out['vib_%s_freq' % sensor] = np.array(vd.get('freq', 0.0), dtype=np.float64)
            if isinstance(env[sensor], np.ndarray):
                out['env_%s_after' % sensor] = env[sensor].copy()           # what the reference left of the caller's array
    if odo_opt is not None:
        out['ref_odo'] = d.ref_odo.data
        out['odo'] = np.stack([d.odo.data[r][k] for r in range(R)])
        out['odo_scale'], out['odo_stdv'] = odo_opt['scale'], odo_opt['stdv']
    if gps:
        out['ref_gps'] = d.ref_gps.data
        out['gps'] = np.stack([d.gps.data[r] for r in range(R)])
        out.update(err_dict_arrays('gps_', imu.gps_err))
    if axis == 9:
        # the WMM field the reference evaluated at the initial position TODAY (geomag.py:23 default date);
        # stored so that the tests feed the same vector (pathgen.py:164-168)
        from gnss_ins_sim.geoparams import geomag
        gm = geomag.GeoMag("WMM.COF")
        f = gm.GeoMag(ini[0] / D2R, ini[1] / D2R, ini[2])
        out['geo_mag_n'] = np.array([f.bx, f.by, f.bz]) / 1000.0
        out['ref_mag'] = d.ref_mag.data
        out['mag'] = np.stack([d.mag.data[r][k] for r in range(R)])
        out.update(err_dict_arrays('mag_', imu.mag_err))
    for ai, a in enumerate(algos):
        key = 'algo%d_' % ai
        out[a + '_att'] = np.stack([d.att_euler.data[key + str(r)][k] for r in range(R)])
        out[a + '_pos'] = np.stack([d.pos.data[key + str(r)][k] for r in range(R)])
        out[a + '_vel'] = np.stack([d.vel.data[key + str(r)][k] for r in range(R)])
    # end-point statistics exactly as Sim.__summary asks for them (ins_sim.py:368-374)
    for dn, ang in (('att_euler', True), ('pos', False), ('vel', False)):
        st = d.get_error_stats(dn, err_stats_start=-1, angle=ang, use_output_units=True)
        for s in ('max', 'avg', 'std'):
            if isinstance(st[s], dict):
                for g, v in st[s].items():
                    out['stat_%s_%s_%s' % (dn, s, g)] = v
            else:
                out['stat_%s_%s_%s' % (dn, s, 'algo0')] = st[s]
    # process-error statistics from t = 2 s (ins_data_manager.py:761-795), internal units, one row per run key
    keys = ['algo%d_%d' % (ai, r) for ai in range(len(algos)) for r in range(R)]
    out['proc_start_s'] = 2.0
    for dn, ang in (('att_euler', True), ('pos', False), ('vel', False)):
        st = d.get_error_stats(dn, err_stats_start=2.0, angle=ang, use_output_units=False)
        for s in ('max', 'avg', 'std'):
            out['proc_%s_%s' % (dn, s)] = np.stack([st[s][kk] for kk in keys])
    if ref_frame == 0:      # extra_opt='ned' (ins_data_manager.py:474-488, 542-552); the error cache must be dropped first
        d._InsDataMgr__err = {}
        st = d.get_error_stats('pos', err_stats_start=-1, angle=False, use_output_units=False, extra_opt='ned')
        for s in ('max', 'avg', 'std'):
            if isinstance(st[s], dict):
                for gname, v in st[s].items():
                    out['ned_end_%s_%s' % (s, gname)] = v
            else:
                out['ned_end_%s_algo0' % s] = st[s]
        st = d.get_error_stats('pos', err_stats_start=2.0, angle=False, use_output_units=False, extra_opt='ned')
        for s in ('max', 'avg', 'std'):
            out['ned_proc_%s' % s] = np.stack([st[s][kk] for kk in keys])
    save(name, **out)


def csv_case():
    """The files Sim.results(data_dir) writes (sim_data.py:117-165 via ins_data_manager.save_data): names, header lines
    and a few rows of every file, from the unmodified reference on a two-run, two-algorithm, rf 0 case with GPS + odo."""
    import tempfile
    csv = MOTION + 'motion_def-90deg_turn.csv'
    odo_opt = {'scale': 0.999, 'stdv': 0.1}
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=True, odo=True, odo_opt=odo_opt)
    ini = read_ini(csv)
    objs = [free_integration.FreeIntegration(ini.copy()), free_integration_odo.FreeIntegration(ini.copy())]
    sim = ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=0, imu=imu, algorithm=objs)
    shim = RandnShim(SEED, 1000, imu.accel_err['b_corr'], imu.gyro_err['b_corr'], gps_m=100, mag=False, odo=True)
    with injected(shim):
        sim.run(2)
    d = tempfile.mkdtemp()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        sim.results(d, err_stats_start=-1)
    names, headers, shapes, rows_kept, values = [], [], [], [], {}
    for f in sorted(os.listdir(d)):
        if not f.endswith('.csv'):
            continue
        with open(os.path.join(d, f)) as fh:
            head = fh.readline().rstrip('\n')
        a = np.atleast_1d(np.genfromtxt(os.path.join(d, f), delimiter=',', skip_header=1))
        if a.ndim == 1:
            a = a[:, None]
        k = rows(a.shape[0], max(1, a.shape[0] // 6))
        names.append(f)
        headers.append(head)
        shapes.append(a.shape)
        values['rows_' + f] = k
        values['data_' + f] = a[k]
    save('csv_files_rf0', seed=SEED, names=np.array(names), headers=np.array(headers), shapes=np.array(shapes),
         ini=ini, odo_scale=odo_opt['scale'], odo_stdv=odo_opt['stdv'], **values)
    print('   csv: %d files: %s' % (len(names), ' '.join(names)))


def summary_case():
    """The text Sim.results() prints (Sim.__summary, ins_sim.py:339-413) on the two-run, two-algorithm, rf 0 case of
    csv_case(), for the three kinds of statistics: end point, process from t = 0 (the reference's default), end point in NED."""
    import contextlib, io
    csv = MOTION + 'motion_def-90deg_turn.csv'
    odo_opt = {'scale': 0.999, 'stdv': 0.1}
    ini = read_ini(csv)
    texts = {}
    for tag, start, opt in (('end', -1, ''), ('process', 0, ''), ('end_ned', -1, 'ned')):
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=True, odo=True, odo_opt=odo_opt)
        objs = [free_integration.FreeIntegration(ini.copy()), free_integration_odo.FreeIntegration(ini.copy())]
        sim = ins_sim.Sim([100.0, 10.0, 0.0], csv, ref_frame=0, imu=imu, algorithm=objs)
        shim = RandnShim(SEED, 1000, imu.accel_err['b_corr'], imu.gyro_err['b_corr'], gps_m=100, mag=False, odo=True)
        with injected(shim):
            sim.run(2)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            sim.results(err_stats_start=start, extra_opt=opt)
        texts[tag] = buf.getvalue()
    save('summary_text_rf0', seed=SEED, ini=ini, odo_scale=odo_opt['scale'], odo_stdv=odo_opt['stdv'],
         **{'text_' + k: np.array(v) for k, v in texts.items()})
    print(texts['end'])


# ------------------------------------------------------------------ T4: the UNPATCHED reference, statistics only
def t4_reference_statistics():
    """SURVEY 8(c) T4.  The reference as shipped -- its own global MT19937 stream, np.random.seed(s) for repeatability --
    on config 1 (90-degree turn @100 Hz, 'mid-accuracy' 6-axis IMU, ref_frame 1, FreeIntegration), R = 1000 runs for each
    of five seeds.  Kept: the end-point errors' mean / std / max per seed and pooled.  No run of the engine can reproduce
    these numbers run by run (different random stream); its 65 536-run statistics must agree within sampling error."""
    csv = MOTION + 'motion_def-90deg_turn.csv'
    ini = read_ini(csv)
    R, seeds = 1000, [2024, 7, 99, 31337, 20260923]
    per_seed = []
    for sd in seeds:
        np.random.seed(sd)
        imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
        sim = ins_sim.Sim([100.0, 0.0, 0.0], csv, ref_frame=1, imu=imu, algorithm=free_integration.FreeIntegration(ini.copy()))
        sim.run(R)
        d = sim.dmgr
        e = np.empty((R, 9))
        for r in range(R):
            k = 'algo0_%d' % r
            a = d.att_euler.data[k][-1] - d.ref_att_euler.data[-1]
            e[r, 0:3] = np.mod(a + np.pi, 2 * np.pi) - np.pi
            e[r, 3:6] = d.pos.data[k][-1] - d.ref_pos.data[-1]
            e[r, 6:9] = d.vel.data[k][-1] - d.ref_vel.data[-1]
        per_seed.append(e)
        print('   T4 seed %d: att std [deg] %s  vel std %s' % (sd, np.degrees(e[:, :3].std(0)), e[:, 6:9].std(0)))
    allr = np.concatenate(per_seed)
    save('t4_c1_reference_stats', seeds=np.array(seeds), runs_per_seed=R,
         mean=np.stack([e.mean(0) for e in per_seed]), std=np.stack([e.std(0) for e in per_seed]),
         maxabs=np.stack([np.abs(e).max(0) for e in per_seed]),
         pooled_mean=allr.mean(0), pooled_std=allr.std(0), pooled_maxabs=np.abs(allr).max(0), pooled_runs=allr.shape[0])


# ------------------------------------------------------------------ truth of a profile that uses every command type + custom mobility
MIXED = """ini lat (deg),ini lon (deg),ini alt (m),ini vx_body (m/s),ini vy_body (m/s),ini vz_body (m/s),ini yaw (deg),ini pitch (deg),ini roll (deg)
31.2,121.4,10,3,0,0,40,0,0
command type,yaw (deg),pitch (deg),roll (deg),vx_body (m/s),vy_body (m/s),vz_body (m/s),command duration (s),GPS visibility
1,0,0,0,0,0,0,4,1
4,75,5,0,2,0,0,12,1
1,0,0,0,0,0,0,3,1
2,10,0,0,8,0,0,15,0
3,-30,-5,10,-3,0,0,10,1
5,20,0,-10,4,0,0,9,1
4,-120,0,0,-1.5,0,0,14,1
1,0,0,0,0,0,0,2,1
"""


def truth_mixed_types():
    """path_gen on a profile that uses command types 1-5 (type 4 = absolute attitude + relative velocity is used by none of
    the reference's own motion files) with a CUSTOM mobility array (Sim(mode=np.array([...])), ins_sim.py:612-640), both
    frames, GPS + odometer."""
    mode = np.array([2.5, 20.0, 60.0])         # m/s^2, deg/s^2, deg/s
    for rf in (0, 1):
        s = ins_sim.Sim([50.0, 5.0, 0.0], MIXED, ref_frame=rf, imu=None, mode=mode)
        ini_pva, motion_def = s._Sim__parse_motion()
        mobility = s._Sim__parse_mode(mode)
        output_def = np.array([[1.0, 50.0], [1.0, 5.0], [1.0, 50.0]])
        r = pathgen.path_gen(ini_pva.copy(), motion_def.copy(), output_def, mobility, ref_frame=rf, magnet=False)
        n, m = r['imu'].shape[0], r['gps'].shape[0]
        k, kg = rows(n, 7), rows(m, 3)
        save('truth_mixed_types_rf%d' % rf, fs=50.0, fs_gps=5.0, n=n, m=m, rows=k, gps_rows=kg, mode=mode, text=np.array(MIXED),
             ini_pva=ini_pva, motion_def=motion_def, mobility=mobility,
             imu=r['imu'][k], nav=r['nav'][k], gps=r['gps'][kg], odo=r['odo'][k])
        print('   mixed rf%d: n = %d, m = %d' % (rf, n, m))


def truth_random_profiles():
    """path_gen on 24 RANDOM motion definitions (seeded): 2-7 segments of command types 1-5 with random rates, speeds,
    durations (some segments of types 2-5 end early, some run out of time), random mobility limits, sample rates and GPS
    visibility, both frames.  Sample counts and sampled rows of every output; the native generator must reproduce them."""
    rng = np.random.RandomState(20260924)
    cases = []
    for i in range(24):
        rf = i % 2
        fs = float(rng.choice([20.0, 50.0, 100.0, 125.0]))
        fs_gps = float(rng.choice([1.0, 5.0, 10.0]))
        ini = np.array([rng.uniform(-60, 60) * D2R, rng.uniform(-170, 170) * D2R, rng.uniform(0, 500),
                        rng.uniform(0, 25), 0.0, 0.0, rng.uniform(-180, 180) * D2R, rng.uniform(-5, 5) * D2R, rng.uniform(-5, 5) * D2R])
        segs = []
        for _ in range(rng.randint(2, 8)):
            t = int(rng.randint(1, 6))
            dur = float(rng.uniform(2.0, 20.0))
            if t == 1:      # rates + accelerations
                row = [t, rng.uniform(-12, 12), rng.uniform(-2, 2), rng.uniform(-4, 4), rng.uniform(-1.5, 1.5), 0.0, 0.0, dur, rng.randint(0, 2)]
            elif t in (2, 4):   # absolute attitude (deg), absolute (2) or relative (4) velocity
                row = [t, rng.uniform(-180, 180), rng.uniform(-8, 8), rng.uniform(-10, 10), rng.uniform(0, 20) if t == 2 else rng.uniform(-4, 4),
                       0.0, 0.0, dur, rng.randint(0, 2)]
            else:           # relative attitude change, absolute (5) or relative (3) velocity
                row = [t, rng.uniform(-90, 90), rng.uniform(-5, 5), rng.uniform(-8, 8), rng.uniform(0, 20) if t == 5 else rng.uniform(-4, 4),
                       0.0, 0.0, dur, rng.randint(0, 2)]
            segs.append(row)
        md = np.array(segs, dtype=np.float64)
        md[:, 1:4] *= D2R
        mob = np.array([rng.uniform(0.5, 4.0), rng.uniform(5.0, 40.0) * D2R, rng.uniform(10.0, 90.0) * D2R])
        output_def = np.array([[1.0, fs], [1.0, fs_gps], [1.0, fs]])
        r = pathgen.path_gen(ini.copy(), md.copy(), output_def, mob.copy(), ref_frame=rf, magnet=False)
        n, m = r['imu'].shape[0], r['gps'].shape[0]
        k, kg = rows(n, max(1, n // 40)), rows(m, max(1, m // 10))
        cases.append(dict(rf=rf, fs=fs, fs_gps=fs_gps, ini=ini, md=md, mob=mob, n=n, m=m, k=k, kg=kg,
                          imu=r['imu'][k], nav=r['nav'][k], gps=r['gps'][kg], odo=r['odo'][k]))
        print('   random %2d: rf %d fs %5.1f segments %d types %s -> n = %d, m = %d' % (i, rf, fs, len(segs), sorted(set(int(x[0]) for x in segs)), n, m))
    out = {'count': len(cases)}
    for i, c in enumerate(cases):
        for key, v in c.items():
            out['c%02d_%s' % (i, key)] = v
    save('truth_random_profiles', **out)


def allan_case():
    n, fs = 360000, 100.0
    x = 0.3 * philox.normal_pair(SEED, 7, 5, np.arange(n, dtype=np.uint64))[0] \
        + 1e-3 * np.cumsum(philox.normal_pair(SEED, 7, 4, np.arange(n, dtype=np.uint64))[1])
    avar, tau = allan.allan_var(x, fs)
    save('allan_ref', seed=SEED, n=n, fs=fs, avar=avar, tau=tau)


def leaves_case():
    """The leaves of the hot path under their reference names, evaluated by the unmodified reference on seeded inputs:
    attitude.euler_update_zyx (attitude.py:679-721; with pitch folds and yaw / roll wraps), pathgen.calc_true_sensor_output
    (pathgen.py:331-411, both frames), pathgen.parse_motion_def (:413-439, the five command types), pathgen.bias_drift
    (:565-594, under the randn shim: the accelerometer drift normals of run 0) and InsDataMgr.array_error (:519-553)."""
    from gnss_ins_sim.attitude import attitude
    from gnss_ins_sim.sim import ins_data_manager
    rs = np.random.RandomState(SEED)
    # --- euler_update_zyx
    N = 240
    x = rs.uniform(-1.0, 1.0, (N, 3)) * np.array([math.pi, 0.5 * math.pi, math.pi])
    w = rs.uniform(-3.0, 3.0, (N, 3))
    dt = rs.choice([0.0025, 0.005, 0.01, 0.02, 0.1], N)
    x[:60, 1] = np.sign(x[:60, 1]) * (0.5 * math.pi - rs.uniform(0.0, 0.02, 60))        # next to the pitch fold
    w[:60, 1:] *= 4.0
    x[60:100, 0] = np.sign(x[60:100, 0]) * (math.pi - rs.uniform(0.0, 0.01, 40))        # next to the yaw / roll wrap
    x[100:140, 2] = np.sign(x[100:140, 2]) * (math.pi - rs.uniform(0.0, 0.01, 40))
    y = np.array([attitude.euler_update_zyx(x[i], w[i], dt[i]) for i in range(N)])
    out = {'eu_x': x, 'eu_w': w, 'eu_dt': dt, 'eu_y': y}
    # --- calc_true_sensor_output
    M = 48
    for rf in (0, 1):
        pos = np.stack([rs.uniform(-1.3, 1.3, M), rs.uniform(-3.1, 3.1, M), rs.uniform(-100.0, 9000.0, M)], 1) if rf == 0 \
            else rs.uniform(-5e3, 5e3, (M, 3))
        vel_b, att = rs.uniform(-40.0, 40.0, (M, 3)), rs.uniform(-1.0, 1.0, (M, 3)) * np.array([math.pi, 1.4, math.pi])
        vdot, adot, g = rs.uniform(-5.0, 5.0, (M, 3)), rs.uniform(-1.0, 1.0, (M, 3)), rs.uniform(9.7, 9.9, M)
        c_nb = np.array([attitude.euler2dcm(att[i], 'zyx').T for i in range(M)])
        res = [pathgen.calc_true_sensor_output(pos[i], vel_b[i], att[i], c_nb[i], vdot[i], adot[i], rf, g[i]) for i in range(M)]
        out.update({'ts%d_pos' % rf: pos, 'ts%d_vel_b' % rf: vel_b, 'ts%d_att' % rf: att, 'ts%d_c_nb' % rf: c_nb, 'ts%d_vdot' % rf: vdot,
                    'ts%d_adot' % rf: adot, 'ts%d_g' % rf: g})
        for k, nm in enumerate(('acc', 'gyro', 'vel_dot_n', 'pos_dot_n')):
            out['ts%d_%s' % (rf, nm)] = np.array([np.asarray(r[k], dtype=np.float64) for r in res])
    # --- parse_motion_def
    seg = rs.uniform(-2.0, 2.0, (25, 9))
    seg[:, 0] = np.tile([1, 2, 3, 4, 5], 5)
    att, vel = rs.uniform(-1.0, 1.0, (25, 3)), rs.uniform(-10.0, 10.0, (25, 3))
    pm = [pathgen.parse_motion_def(seg[i], att[i], vel[i]) for i in range(25)]
    out.update({'pm_seg': seg, 'pm_att': att, 'pm_vel': vel, 'pm_att_com': np.array([np.asarray(p[0], dtype=np.float64) for p in pm]),
                'pm_vel_com': np.array([np.asarray(p[1], dtype=np.float64) for p in pm])})
    # --- bias_drift: the drift normals of the accelerometer of run 0 in the reference's own call order
    n, fs = 3000, 100.0
    corr, drift = np.array([100.0, np.inf, 0.5]), np.array([3e-4, 2e-4, 5e-4])
    with injected(RandnShim(SEED, n, corr, corr)):
        bd = pathgen.bias_drift(corr, drift, n, fs)
    out.update({'bd_seed': SEED, 'bd_n': n, 'bd_fs': fs, 'bd_corr': corr, 'bd_drift': drift, 'bd_out': bd})
    # --- array_error
    mgr = ins_data_manager.InsDataMgr([100.0, 0.0, 0.0], 0)
    ang_x, ang_r = rs.uniform(-7.0, 7.0, (50, 3)), rs.uniform(-7.0, 7.0, (50, 3))
    lla_r = np.stack([rs.uniform(-1.3, 1.3, 50), rs.uniform(-3.1, 3.1, 50), rs.uniform(0.0, 3000.0, 50)], 1)
    lla_x = lla_r + rs.uniform(-1.0, 1.0, (50, 3)) * np.array([1e-5, 1e-5, 30.0])
    out.update({'ae_ang_x': ang_x, 'ae_ang_r': ang_r, 'ae_ang': mgr.array_error(ang_x, ang_r, angle=True),
                'ae_lla_x': lla_x, 'ae_lla_r': lla_r, 'ae_ned': mgr.array_error(lla_x, lla_r, lla=1),
                'ae_ecef': mgr.array_error(lla_x, lla_r, lla=2)})
    save('leaves', **out)


def emit_profiles():
    prof = os.path.join(REPO, 'gnss-ins-sim_amd', 'motion_profiles')
    emit_profile(MOTION + 'motion_def-90deg_turn.csv', prof + '/turn_90deg.csv', '90-degree turn, 10 s')
    emit_profile(MOTION + 'motion_def-long_drive.csv', prof + '/long_drive.csv', 'long drive, <=1410 s')
    emit_profile(MOTION + 'motion_def-Allan.csv', prof + '/static_1800s.csv', 'static, 1800 s')


def t3_white():
    white = {k: v for k, v in DEMO_IMU.items() if not k.endswith('_corr')}
    white['gyro_b'] = np.array([10.0, -20.0, 30.0])
    white['accel_b'] = np.array([1e-3, -2e-3, 3e-3])
    t3_case('t3_white_gps_rf0', 0, white, True, {'scale': 1.001, 'stdv': 0.05}, ['fi', 'odo'], 3, fs_gps=10.0)


def t3_mag9(rf):
    mag9 = dict(DEMO_IMU)
    mag9.update({'mag_si': np.array([[1.02, 0.01, -0.02], [0.03, 0.97, 0.01], [-0.01, 0.02, 1.05]]),
                 'mag_hi': np.array([5.0, -8.0, 12.0]), 'mag_std': np.array([0.2, 0.1, 0.3])})
    t3_case('t3_mag9_gps_rf%d' % rf, rf, dict(mag9), True, None, ['fi'], 2, fs_gps=10.0, axis=9)


# Every case runs in an interpreter of its own.  The reference's IMU(accuracy=dict) ALIASES the module-level
# 'low-accuracy' dicts and then overwrites them with the caller's values (gnss_ins_sim/sim/imu_model.py:110-112, 138-158),
# so in one interpreter every 'low-accuracy' case that follows a dict case silently gets the dict's IMU -- round 2's
# t3_low_rf1 was the demo IMU under another name.  One process per case makes the recipe independent of the order.
CASES = [
    ('profiles', 'emit_profiles()', ['motion_profiles/turn_90deg.csv', 'motion_profiles/long_drive.csv', 'motion_profiles/static_1800s.csv']),
    ('t1_tumble', 't1_tumble()', ['t1_fixture_tumble.npz']),
    ('t1_rates', 't1_rates()', ['t1_rates.npz']),
    ('t1_bosch', "t1_fixture('bosch')", ['t1_fixture_bosch.npz']),
    ('t1_nxp', "t1_fixture('nxp')", ['t1_fixture_nxp.npz']),
    ('t2_turn_rf1', 't2_turn(1)', ['t2_turn_rf1.npz']),
    ('t2_turn_rf0', 't2_turn(0)', ['t2_turn_rf0.npz']),
    ('t3_demo_rf1', "t3_case('t3_demo_rf1', 1, dict(DEMO_IMU), False, {'scale': 0.999, 'stdv': 0.1}, ['odo', 'fi'], 4)", ['t3_demo_rf1.npz']),
    ('t3_mid_rf0', "t3_case('t3_mid_rf0', 0, 'mid-accuracy', False, None, ['fi'], 4)", ['t3_mid_rf0.npz']),
    ('t3_low_rf1', "t3_case('t3_low_rf1', 1, 'low-accuracy', False, None, ['fi'], 3)", ['t3_low_rf1.npz']),
    ('t3_high_odo_rf0', "t3_case('t3_high_odo_rf0', 0, 'high-accuracy', False, {'scale': 1.002, 'stdv': 0.02}, ['odo', 'fi'], 3)", ['t3_high_odo_rf0.npz']),
    ('t3_white_gps_rf0', 't3_white()', ['t3_white_gps_rf0.npz']),
    ('t3_mag9_gps_rf0', 't3_mag9(0)', ['t3_mag9_gps_rf0.npz']),
    ('t3_mag9_gps_rf1', 't3_mag9(1)', ['t3_mag9_gps_rf1.npz']),
    ('t3_drive200_rf0', "t3_case('t3_drive200_rf0', 0, 'low-accuracy', True, {'scale': 0.998, 'stdv': 0.05}, ['fi', 'odo'], 2, fs_gps=5.0, csv=DRIVE, fs=200.0)", ['t3_drive200_rf0.npz']),
    ('t3_vib_random_rf1', "t3_case('t3_vib_random_rf1', 1, 'mid-accuracy', False, None, ['fi'], 3, env={'acc': '[0.03 0.001 0.01]-random', 'gyro': '[0.1 0.2 0.3]d-random'})", ['t3_vib_random_rf1.npz']),
    ('t3_vib_sin_rf0', "t3_case('t3_vib_sin_rf0', 0, 'low-accuracy', False, {'scale': 0.999, 'stdv': 0.1}, ['fi', 'odo'], 3, env={'acc': '[0.01 0.02 0.03]g-2.5Hz-sinusoidal', 'gyro': '[0.5 0.4 0.3]d-0.7Hz-sinusoidal'})", ['t3_vib_sin_rf0.npz']),
    ('t3_vib_mixed_rf1', "t3_case('t3_vib_mixed_rf1', 1, dict(DEMO_IMU), False, None, ['fi'], 2, env={'acc': '[0.5 0.1 0.2]-12Hz-sinusoidal', 'gyro': '[0.002 0.001 0.003]-random'})", ['t3_vib_mixed_rf1.npz']),
    ('t3_vib_psd_rf1', "t3_case('t3_vib_psd_rf1', 1, 'mid-accuracy', False, None, ['fi'], 3, env={'acc': psd_env('coarse'), 'gyro': psd_env('grid501', 1e-3)})", ['t3_vib_psd_rf1.npz']),
    ('t3_vib_psd_odd_rf0', "t3_case('t3_vib_psd_odd_rf0', 0, 'low-accuracy', False, {'scale': 0.999, 'stdv': 0.1}, ['fi', 'odo'], 2, fs_gps=10.0, csv=PSD_ODD, env={'acc': '[0.02 0.01 0.03]-random', 'gyro': psd_env('coarse', 1e-2)})", ['t3_vib_psd_odd_rf0.npz']),
    ('t3_vib_psd_tiled_rf0', "t3_case('t3_vib_psd_tiled_rf0', 0, 'high-accuracy', False, None, ['fi'], 2, fs_gps=10.0, csv=PSD_LONG, env={'acc': psd_env('coarse'), 'gyro': psd_env('coarse', 1e-3)})", ['t3_vib_psd_tiled_rf0.npz']),
    ('csv_case', 'csv_case()', ['csv_files_rf0.npz']),
    ('summary_case', 'summary_case()', ['summary_text_rf0.npz']),
    ('allan_case', 'allan_case()', ['allan_ref.npz']),
    ('t2_long_drive', 't2_long_drive()', ['t2_long_drive_rf0.npz']),
    ('truth_mixed_types', 'truth_mixed_types()', ['truth_mixed_types_rf0.npz', 'truth_mixed_types_rf1.npz']),
    ('truth_random_profiles', 'truth_random_profiles()', ['truth_random_profiles.npz']),
    ('t4_reference_statistics', 't4_reference_statistics()', ['t4_c1_reference_stats.npz']),
    ('leaves', 'leaves_case()', ['leaves.npz']),
]


def run_cases(names, out_dir=None):
    """One fresh interpreter per case (see CASES).  out_dir: write there instead of tests/golden/ (and motion_profiles/)."""
    import subprocess
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', MPLBACKEND='Agg')
    if out_dir:
        env['GINSIM_GOLDEN_OUT'] = out_dir
    for name, _, _ in CASES:
        if names and name not in names:
            continue
        print('== %s' % name, flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--case', name], env=env)


def check():
    """Regenerate every golden into a scratch directory and byte-compare with the committed files."""
    import filecmp
    import tempfile
    tmp = tempfile.mkdtemp(prefix='golden_check_')
    run_cases(sys.argv[2:], tmp)
    prof = os.path.join(REPO, 'gnss-ins-sim_amd')
    bad = []
    for name, _, files in CASES:
        if sys.argv[2:] and name not in sys.argv[2:]:
            continue
        for f in files:
            committed = os.path.join(prof, f) if f.startswith('motion_profiles/') else os.path.join(HERE, f)
            fresh = os.path.join(tmp, f)
            same = os.path.exists(fresh) and os.path.exists(committed) and filecmp.cmp(fresh, committed, shallow=False)
            note = 'identical' if same else 'DIFFERS'
            if not same and f.endswith('.npz') and os.path.exists(fresh) and os.path.exists(committed):
                # The 9-axis cases hold the geomagnetic field the reference's WMM gave on the day they were made (geomag.py:23: the
                # date defaults to date.today(); the tests feed the STORED vector).  Regenerated on another day that vector and the
                # magnetometer rows made from it move, nothing else may.
                a, b = np.load(committed), np.load(fresh)
                moved = sorted(k for k in set(a.files) | set(b.files)
                               if k not in a.files or k not in b.files or a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]))
                if moved and all(k == 'geo_mag_n' or k.startswith(('mag', 'ref_mag')) for k in moved) and 'geo_mag_n' in moved:
                    same, note = True, 'identical but for the WMM field of the day (%s)' % ', '.join(moved)
                else:
                    note = 'DIFFERS in ' + ', '.join(moved[:8])
            print('%-40s %s' % (f, note))
            if not same:
                bad.append(f)
    print('%d file(s) differ' % len(bad) if bad else 'all golden files reproduce bit-identically')
    return 1 if bad else 0


if __name__ == '__main__':
    # make_golden.py                 every case, each in a fresh interpreter
    # make_golden.py NAME [NAME..]   the named cases (names of CASES), each in a fresh interpreter
    # make_golden.py --check [NAME..] regenerate into a scratch directory and byte-compare with the committed files
    # make_golden.py --case NAME     (internal) run ONE case in this interpreter
    if len(sys.argv) > 2 and sys.argv[1] == '--case':
        code = dict((c[0], c[1]) for c in CASES)[sys.argv[2]]
        eval(code)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == '--check':
        sys.exit(check())
    run_cases(sys.argv[1:])
