#!/usr/bin/env python3
"""The Allan call's time per ALLOCATION: several hipMalloc'ed buffers of the series' exact size in one fresh process."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim._lib import lib, check

S, n, fs = 192, 1440000, 400.0
ctx = ginsim.Context(0)
ctx.pool_limit = 0                      # every free is a hipFree
rng = np.random.default_rng(0)
host = rng.normal(size=(S, n))

def timed(buf, tag):
    for _ in range(30):
        ginsim.allan_var(ctx, buf, n, S, n, fs)
    ts = []
    for _ in range(20):
        ctx.timer_begin(); ginsim.allan_var(ctx, buf, n, S, n, fs); ts.append(ctx.timer_end())
    print('%-40s 0x%x  %.4f ms avg  %.4f min' % (tag, buf.ptr, sum(ts) / len(ts), min(ts)), flush=True)

bufs = []
for k in range(6):
    b = ctx.malloc(host.nbytes + int(os.environ.get('EXTRA', 0)))
    check(lib.ginsim_memcpy_h2d(ctx.handle, b.ptr, host.ctypes.data, host.nbytes))
    bufs.append(b)
    timed(b, 'allocation %d' % k)
for b in bufs:
    b.free()
for k in range(3):
    b = ctx.malloc(host.nbytes)
    check(lib.ginsim_memcpy_h2d(ctx.handle, b.ptr, host.ctypes.data, host.nbytes))
    timed(b, 'after freeing all: allocation %d' % k)
    bufs.append(b)
