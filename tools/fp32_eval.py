"""fp32 kernel: deviation from fp64/reference and timing (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO, os.path.join(REPO, 'tests')]
import numpy as np
import ginsim
from ginsim import workloads
from conftest import load_golden
ctx = ginsim.Context(0)
zero = {'b': np.zeros(3), 'b_drift': np.zeros(3), 'b_corr': np.full(3, 100.0), 'arw': np.zeros(3), 'vrw': np.zeros(3)}
for rf in (1, 0):
    g = load_golden('t2_turn_rf%d' % rf)
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, zero, zero, ini, runs=2, algos=('free', 'odo'), odo_err={'scale': 1.0, 'stdv': 0.0},
                               seed=1, keep_traj=True, precision='f32').run()
    k = g['rows']
    for a, tag in (('free', 'fi'), ('odo', 'odo')):
        att, pos, vel = job.trajectories(a, [1])
        da = np.abs(np.mod(att[0][k] - g[tag + '_att'] + np.pi, 2 * np.pi) - np.pi).max(0)
        print('rf=%d %s noise-free |d att| %s  |d pos| %s  |d vel| %s' % (rf, a, da, np.abs(pos[0][k] - g[tag + '_pos']).max(0), np.abs(vel[0][k] - g[tag + '_vel']).max(0)))
    print('   end errors f32', job.end_errors('free')[0])
    job.release()
acc, gyr = workloads.imu_grade('mid-accuracy')
for rf in (1, 0):
    ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, rf)
    st = {}
    for prec in ('f64', 'f32'):
        job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, ini, runs=65536, seed=5, precision=prec).run()
        st[prec] = job.stats('free'); job.release()
    print('rf=%d std ratio f32/f64' % rf, st['f32'].std / st['f64'].std)
    print('      mean f32', st['f32'].mean, '\n      mean f64', st['f64'].mean, '\n      sigma/sqrtR', st['f64'].std / 256)
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
for R in (65536, 262144):
    for keep in (False, True):
        for prec in ('f64', 'f32'):
            job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=5, keep_sensors=keep, keep_traj=keep, precision=prec).run()
            ts = []
            for _ in range(4):
                ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
            ms = min(ts)
            print('R=%d keep=%d %s: %.3f ms  %.3e sample.MC/s  %.0f GB/s' % (R, keep, prec, ms, R * 1000 / ms * 1e3, job.bytes_written() / ms / 1e6))
            job.release()
