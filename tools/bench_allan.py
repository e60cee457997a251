#!/usr/bin/env python3
"""Allan-variance kernel benchmark (BASELINE config 5: 3600 s static @ 400 Hz, n = 1 440 000 per series).
Prints one JSON line: series/s, achieved HBM GB/s against the 8 B/sample algorithmic model."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim

S = int(os.environ.get('SERIES', 192)); n = int(os.environ.get('N', 1440000)); fs = 400.0
ctx = ginsim.Context(0)
rng = np.random.default_rng(0)
x = rng.normal(size=(S, n))
buf = ctx.upload(x)
for _ in range(int(os.environ.get('WARM', 40))):       # the call settles after ~30 back-to-back calls (clock / power management)
    ginsim.allan_var(ctx, buf, n, S, n, fs)
ts = []
for _ in range(30):
    ctx.timer_begin(); t0 = time.perf_counter()
    avar, tau = ginsim.allan_var(ctx, buf, n, S, n, fs)
    ts.append((ctx.timer_end(), (time.perf_counter() - t0) * 1e3))
ms = sum(t[0] for t in ts) / len(ts)
alg = 8.0 * S * n
print(json.dumps({'kernel': 'ginsim_allan: %d decade levels' % int(np.ceil(np.log10(n // 9))), 'ms_min': min(t[0] for t in ts), 'series': S, 'n': n, 'ntau': int(tau.size),
                  'ms': ms, 'ms_wall': min(t[1] for t in ts), 'algorithmic_bytes': alg, 'achieved_GBps': alg / ms / 1e6,
                  'frac_of_8TBps': alg / ms / 1e6 / 8000.0, 'samples_per_s': S * n / ms * 1e3}))
if os.environ.get('CPU'):
    from oracle import c_oracle
    t0 = time.perf_counter(); c_oracle.allan_var(x[0], fs); dt = time.perf_counter() - t0
    print(json.dumps({'cpu_oracle_one_series_s': dt, 'cpu_samples_per_s': n / dt}))
