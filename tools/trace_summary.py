#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace result database: python tools/trace_summary.py <dir or .db> [name filter]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
con = sqlite3.connect(path)
print('%-90s %8s %6s %10s %10s %10s' % ('kernel', 'grid', 'calls', 'avg_us', 'min_us', 'max_us'))
for name, gx, gy, cnt, avg, mn, mx in con.execute(
        "select name, grid_x, grid_y, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
        "group by name, grid_x, grid_y order by sum(end-start) desc"):
    if flt in name:
        print('%-90s %8s %6d %10.1f %10.1f %10.1f' % (name[:90], '%dx%d' % (gx, gy), cnt, avg / 1e3, mn / 1e3, mx / 1e3))
