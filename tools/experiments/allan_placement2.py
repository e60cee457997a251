#!/usr/bin/env python3
"""The Allan call's time against the ADDRESS of its input: one 6 GB allocation, the 192 x 1 440 000 series copied to different
offsets inside it and with different series strides."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim._lib import lib, check

S, n, fs = 192, 1440000, 400.0
ctx = ginsim.Context(0)
rng = np.random.default_rng(0)
host = rng.normal(size=(S, n))
big = ctx.malloc(6 * 2 ** 30)
print('base 0x%x' % big.ptr, flush=True)

def put(off, stride):
    if stride == n:
        check(lib.ginsim_memcpy_h2d(ctx.handle, big.ptr + off, host.ctypes.data, host.nbytes))
    else:
        pad = np.zeros((S, stride))
        pad[:, :n] = host
        check(lib.ginsim_memcpy_h2d(ctx.handle, big.ptr + off, pad.ctypes.data, pad.nbytes))

def timed(off, stride):
    put(off, stride)
    p = big.ptr + off
    for _ in range(30):
        ginsim.allan_var(ctx, p, n, S, stride, fs)
    ts = []
    for _ in range(20):
        ctx.timer_begin(); ginsim.allan_var(ctx, p, n, S, stride, fs); ts.append(ctx.timer_end())
    print('offset %10d (0x%x mod 2MiB = %7d)  stride n+%-6d  %.4f ms avg  %.4f min' % (off, p, p % (2 << 20), stride - n, sum(ts) / len(ts), min(ts)), flush=True)

for off in (0, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 1 << 30, (1 << 30) + 12288):
    timed(off, n)
for stride in (n + 2, n + 16, n + 512, n + 2048, n + 2560, n + 4096, n + 65536 // 8, 1441792):
    timed(0, stride)
a = ctx.upload(host)
print('separate upload at 0x%x (mod 2MiB %d)' % (a.ptr, a.ptr % (2 << 20)))
