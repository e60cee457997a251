#!/usr/bin/env python3
"""A/B of two builds of libginsim.so on BASELINE config 3's launch (long_drive @200 Hz, ref_frame 0, 262 144 runs, statistics only,
online process statistics): each library in fresh processes, alternating, on ONE box.

    python tools/experiments/c3_ab.py gnss-ins-sim_amd/lib/libginsim.so gnss-ins-sim_amd/lib/libginsim_x.so [rounds]

A library name followed by ":shifted" runs with proc_plain_sums forced to 0 (the sums shifted about the launch's initial error, what
a job that starts off the truth gets); without it the job decides (C3 starts on the truth: plain sums).
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

CHILD = r'''
import json, sys
sys.path.insert(0, %r)
import ginsim
from ginsim import workloads
ctx = ginsim.Context(0)
ini, truth, _ = workloads.truth_from_profile('long_drive', 200.0, 0, fs_gps=10.0, gps=True)
acc, gyr = workloads.imu_grade('mid-accuracy')
job = ginsim.MonteCarloJob(ctx, 200.0, 0, truth, acc, gyr, ini, runs=262144, seed=7, keep_sensors=False, keep_traj=False,
                           proc_first=0, end_ned=True)
import os
if os.environ.get('C3_AB_SHIFTED'):
    job.params.proc_plain_sums = 0
job.run()
ms = []
for i in range(3):
    ctx.event_record(2 * i)
    job.launch()
    ctx.event_record(2 * i + 1)
ms = [ctx.event_elapsed(2 * i, 2 * i + 1) for i in range(3)]
print(json.dumps({'kernel': job.kernel_name(), 'ms': ms}))
''' % os.path.join(REPO, 'gnss-ins-sim_amd')

if __name__ == '__main__':
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    out = {l: [] for l in libs}
    for _ in range(rounds):
        for l in libs:
            env = dict(os.environ, GINSIM_LIB=os.path.abspath(l.split(':')[0]))
            env.pop('C3_AB_SHIFTED', None)
            if l.endswith(':shifted'):
                env['C3_AB_SHIFTED'] = '1'
            p = subprocess.run([sys.executable, '-c', CHILD], env=env, stdout=subprocess.PIPE, universal_newlines=True, check=True)
            out[l].append(json.loads(p.stdout.strip().splitlines()[-1]))
    print(json.dumps(out, indent=1))
