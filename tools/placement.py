"""Wave placement / imbalance probe for mc_kernel (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'gnss-ins-sim_amd'), REPO]
if os.environ.get('WITH_TORCH'):
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
import numpy as np
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('turn_90deg', 100.0, 1)
acc, gyr = workloads.imu_grade('mid-accuracy')
R = int(os.environ.get('RUNS', 65536))
for tb in (64, 128, 256):
    for keep in (True, False):
        job = ginsim.MonteCarloJob(ctx, 100.0, 1, truth, acc, gyr, ini, runs=R, seed=1, keep_sensors=keep, keep_traj=keep)
        nw = (R + 63) // 64
        tr = ctx.malloc(nw * 32)
        job.params.wave_trace = tr.ptr
        job.params.block_threads = tb
        job.run()
        ts = []
        for _ in range(5):
            ctx.timer_begin(); job.launch(); ts.append(ctx.timer_end())
        t = ctx.download(tr, (nw, 4), dtype=np.uint64)
        hw, xcc = t[:, 0], t[:, 1] & 0xF
        simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        slot = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
        u, c = np.unique(slot, return_counts=True)
        cus = np.unique(slot // 4).size
        dur = (t[:, 3] - t[:, 2]).astype(np.float64)
        span = float(t[:, 3].max() - t[:, 2].min())
        print('tb=%3d keep=%d  ms min/med/max %.3f %.3f %.3f | SIMDs used %d (CUs %d) waves/SIMD hist %s | wave dur mean/max %.3g/%.3g of span %.3g' % (
            tb, keep, min(ts), np.median(ts), max(ts), u.size, cus, dict(zip(*np.unique(c, return_counts=True))), dur.mean(), dur.max(), span), flush=True)
        job.release(); tr.free()
