"""MonteCarloJob.spread_outputs on the headline job, in a process that first holds PRE_GB of the device memory (moves where the
job's own regions land): what the search finds and how long it takes."""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, 'gnss-ins-sim_amd'), ROOT]
import ginsim
from ginsim import workloads
import bench

ctx = ginsim.Context(0)
pre = [ctx.malloc(int(g) << 30) for g in os.environ.get('PRE_GB', '').split(',') if g]
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
acc, gyr = workloads.imu_grade('mid-accuracy')
job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=65536, algos=('free',), seed=bench.SEED, keep_sensors=True, keep_traj=True)
job.run()
t0 = time.perf_counter()
r = job.spread_outputs()
dt = time.perf_counter() - t0
for _ in range(30):
    job.launch()
ctx.sync()
ms, mn = bench.time_launches(ctx, job.launch, 40, warm=0)
print(json.dumps({'pre_gb': os.environ.get('PRE_GB', ''), 'search_s': round(dt, 2), 'after_ms': round(ms, 4), 'frac': round(job.bytes_written() / (ms * 1e-3) / 8e12, 3),
                  'search': {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, 'free_gb_after': round(ctx.mem_info()[0] / 2 ** 30, 1)}))
