#!/usr/bin/env python3
"""Why is the Allan call 5 % slower inside bench.py than in a process of its own?  Hypothesis: where the 2.2 GB of series land.
Times the call on (A) a buffer allocated in a fresh process, (B) a buffer allocated after the legs of the bench have left GBs of
freed regions parked in the context's pool, (C) the same after the pool was given back to the driver."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads

S, n, fs = 192, 1440000, 400.0
ctx = ginsim.Context(0)
rng = np.random.default_rng(0)
host = rng.normal(size=(S, n))

def timed(buf, tag):
    for _ in range(40):
        ginsim.allan_var(ctx, buf, n, S, n, fs)
    ts = []
    for _ in range(30):
        ctx.timer_begin(); ginsim.allan_var(ctx, buf, n, S, n, fs); ts.append(ctx.timer_end())
    print('%-52s %.4f ms avg  %.4f min  (%.3f of 8 TB/s)' % (tag, sum(ts) / len(ts), min(ts), 8.0 * S * n / (sum(ts) / len(ts)) / 1e6 / 8000), flush=True)

a = ctx.upload(host)
timed(a, 'A fresh process')
# what the legs of the bench do: multi-GB jobs come and go, their regions stay in the pool
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
acc, gyr = workloads.imu_grade('mid-accuracy')
for runs in (65536, 131072, 262144):
    for prec in ('f64', 'f32'):
        j = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=runs, seed=1, keep_sensors=True, keep_traj=True, precision=prec).run()
        j.release()
print('pool holds %.1f GB' % (ctx._pool_bytes / 1e9), flush=True)
b = ctx.malloc(8 * S * n + 4096)            # a size the pool does not have: a fresh hipMalloc among the parked regions
from ginsim._lib import lib, check
check(lib.ginsim_memcpy_h2d(ctx.handle, b.ptr, host.ctypes.data, host.nbytes))
timed(b, 'B allocated with the pool full')
timed(a, 'A again')
ctx.release_pool()
c = ctx.malloc(8 * S * n + 8192)
check(lib.ginsim_memcpy_h2d(ctx.handle, c.ptr, host.ctypes.data, host.nbytes))
timed(c, 'C allocated after release_pool()')
timed(b, 'B again')
