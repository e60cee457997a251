import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np, ginsim
S, n, fs = 192, 1440000, 400.0
ctx = ginsim.Context(0)
x = np.random.default_rng(0).normal(size=(S, n))
buf = ctx.upload(x)
ginsim.allan_var(ctx, buf, n, S, n, fs)
ts = []
for _ in range(120):
    ctx.timer_begin(); ginsim.allan_var(ctx, buf, n, S, n, fs); ts.append(ctx.timer_end())
print(' '.join('%.3f' % t for t in ts))
print('first 10 avg %.4f, last 60 avg %.4f min %.4f' % (sum(ts[:10]) / 10, sum(ts[60:]) / 60, min(ts)))
