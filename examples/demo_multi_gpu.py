#!/usr/bin/env python3
"""One process, every GPU of the node: the Monte-Carlo runs of a Sim are spread over the visible devices by contiguous global
run ranges (one context and one host thread per device, no torch, no launcher) and the per-device statistics records are folded
with the library's Chan merge.  The reference's loop being sharded: gnss_ins_sim/sim/ins_sim.py:490-506.

    PYTHONPATH=gnss-ins-sim_amd python examples/demo_multi_gpu.py [runs] [devices]     # devices: all | 0,1,2 | 0,0 (two contexts on GPU 0)

An UNCHANGED script gets the same without the keyword: set GINSIM_DEVICES=all, or run a batch of at least 2^30 sample x run
products (BASELINE configs 3 and 4), which an unconfigured Sim spreads by itself.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd'))

from gnss_ins_sim.sim import imu_model, ins_sim                      # noqa: E402
from demo_algorithms import free_integration                         # noqa: E402

MOTION = os.path.join(os.path.dirname(HERE), 'gnss-ins-sim_amd', 'motion_profiles', 'turn_90deg.csv')


def main(runs, devices):
    imu = imu_model.IMU(accuracy='mid-accuracy', axis=6, gps=False)
    ini = np.genfromtxt(MOTION, delimiter=',', skip_header=1, max_rows=1)
    ini[0:2] *= np.pi / 180
    ini[6:9] *= np.pi / 180
    for dev in (None, devices):
        sim = ins_sim.Sim([100.0, 0.0, 0.0], MOTION, ref_frame=1, imu=imu, algorithm=free_integration.FreeIntegration(ini), seed=1,
                          device=0 if dev is None else None, devices=dev)
        t0 = time.perf_counter()
        sim.run(runs)
        dt = time.perf_counter() - t0
        sim.results(err_stats_start=-1)
        print('devices %s: %d runs in %.4f s; run %d of the batch: final velocity %s' % (
            'one (GPU 0)' if dev is None else sim.mc.devices, runs, dt, runs - 1, sim.dmgr.vel.data['algo0_%d' % (runs - 1)][-1]))


if __name__ == '__main__':
    d = sys.argv[2] if len(sys.argv) > 2 else 'all'
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 262144, d if d == 'all' else [int(x) for x in d.split(',')])
