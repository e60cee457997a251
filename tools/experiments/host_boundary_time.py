"""PCIe-inclusive rate of the host-buffer boundary ginsim_free_integration (the plugin's run(set_of_input)) -- development aid."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
n = truth['ref_gyro'].shape[0]
rng = np.random.default_rng(0)
for R in (1, 64, 4096, 65536):
    g = np.broadcast_to(truth['ref_gyro'], (R, n, 3)) + 1e-4 * rng.normal(size=(R, n, 3))
    a = np.broadcast_to(truth['ref_accel'], (R, n, 3)) + 1e-3 * rng.normal(size=(R, n, 3))
    ginsim.free_integration_host(ctx, 'free', 1, 100.0, g, a, ini=ini)
    ts = []
    for _ in range(3 if R > 4096 else 8):
        t0 = time.perf_counter(); ginsim.free_integration_host(ctx, 'free', 1, 100.0, g, a, ini=ini); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print('R=%6d n=%d: %.3f ms per call, %.3g sample*run/s, host traffic %.1f MB -> %.2f GB/s PCIe-inclusive (pageable NumPy arrays)'
          % (R, n, t * 1e3, R * n / t, R * n * 120 / 1e6, R * n * 120 / t / 1e9))
    # the same with every host buffer page-locked (ginsim.pinned_empty)
    gp, ap = ginsim.pinned_empty(ctx, g.shape), ginsim.pinned_empty(ctx, a.shape)
    gp[...] = g
    ap[...] = a
    r0 = ginsim.free_integration_host(ctx, 'free', 1, 100.0, g, a, ini=ini)
    outp = tuple(ginsim.pinned_empty(ctx, (R, n, 3)) for _ in range(3))       # kept across calls
    ts = []
    for _ in range(3 if R > 4096 else 8):
        t0 = time.perf_counter(); r1 = ginsim.free_integration_host(ctx, 'free', 1, 100.0, gp, ap, ini=ini, out=outp); ts.append(time.perf_counter() - t0)
    assert all((x == y).all() for x, y in zip(r0, r1))
    t = min(ts)
    print('          pinned  : %.3f ms per call, %.3g sample*run/s -> %.2f GB/s' % (t * 1e3, R * n / t, R * n * 120 / t / 1e9))
    del gp, ap, r1, outp
