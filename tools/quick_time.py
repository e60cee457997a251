"""Ad-hoc kernel timing on the GPU box (development aid, not the bench)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO, os.path.join(REPO, 'tests')]
import numpy as np
import ginsim
from conftest import load_golden

ctx = ginsim.Context(0)
print(ctx.name())
t2 = load_golden('t2_turn_rf1')
D2R = np.pi / 180
gyr = {'b': np.zeros(3), 'b_drift': np.full(3, 3.5) * D2R / 3600, 'b_corr': np.full(3, 100.0), 'arw': np.full(3, 0.25) * D2R / 60}
acc = {'b': np.zeros(3), 'b_drift': np.full(3, 5e-5), 'b_corr': np.full(3, 100.0), 'vrw': np.full(3, 0.03) / 60}
for rf in (1, 0):
    g = load_golden('t2_turn_rf%d' % rf)
    r = ginsim.pathgen(g['ini_pva'], g['motion_def'], 100.0, 10.0, g['mobility'], rf)
    truth = {'ref_accel': r['imu'][:, 1:4], 'ref_gyro': r['imu'][:, 4:7], 'ref_pos': r['nav'][:, 1:4],
             'ref_vel': r['nav'][:, 4:7], 'ref_att': r['nav'][:, 7:10], 'ref_odo': r['odo'][:, 2]}
    for R in [int(x) for x in os.environ.get('RUNS', '65536,262144').split(',')]:
        for keep in (False, True):
            for algos in (('free',), ('free', 'odo')):
                if keep and R * 1000 * 8 * (6 + 9 * len(algos)) > 60e9:
                    continue
                job = ginsim.MonteCarloJob(ctx, 100.0, rf, truth, acc, gyr, g['ini_pva'], runs=R, algos=algos,
                                           odo_err={'scale': 0.999, 'stdv': 0.1} if 'odo' in algos else None, seed=1, keep_sensors=keep, keep_traj=keep)
                job.run()
                ts = []
                for _ in range(3):
                    ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
                ms = min(ts)
                print('rf=%d R=%7d keep=%d algos=%-12s %8.3f ms  %.3e sample.MC/s  %.1f GB/s written' % (
                    rf, R, keep, '+'.join(algos), ms, R * 1000 / ms * 1e3, job.bytes_written() / ms / 1e6), flush=True)
                job.release()
