"""Replicate bench.py's step sequence with wave tracing (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
acc, gyr = workloads.imu_grade('mid-accuracy')
R = 65536
nw = R // 64
def hist(t):
    hw, xcc = t[:, 0], t[:, 1] & 0xF
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    slot = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    u, c = np.unique(slot, return_counts=True)
    return dict(zip(*[x.tolist() for x in np.unique(c, return_counts=True)]))
for variant in ('plain', 'events+stats', 'stats'):
    job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=True, keep_traj=True)
    tr = ctx.malloc(nw * 32)
    job.params.wave_trace = tr.ptr
    job.run()
    out = []
    for s in range(6):
        if variant == 'offset':
            job.params.run_offset = s * R
        if 'events' in variant:
            ctx.event_record(2 * s)
        else:
            ctx.timer_begin()
        job.launch()
        if 'events' in variant:
            ctx.event_record(2 * s + 1)
            ms = None
        else:
            ms = ctx.timer_end()
        if 'stats' in variant:
            job.stats('free')
        if variant == 'sync-only':
            ctx.sync()
        if ms is None:
            ms = ctx.event_elapsed(2 * s, 2 * s + 1)
        h = hist(ctx.download(tr, (nw, 4), dtype=np.uint64))
        out.append('%.2f %s' % (ms, h))
    print(variant, '|', ' ; '.join(out), flush=True)
    job.release(); tr.free()
