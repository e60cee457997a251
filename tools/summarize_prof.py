#!/usr/bin/env python3
"""Turn the rocprofv3 result databases under gpurun_out/prof_* into the small text summaries committed under
profiles/ (kernel trace stats, PMC counters, HBM traffic per launch)."""
import csv
import re
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, 'gpurun_out')
DST = os.path.join(REPO, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'


def db(name):
    p = os.path.join(SRC, name, 'bench_results.db')
    return sqlite3.connect(p) if os.path.exists(p) else None


os.makedirs(DST, exist_ok=True)
con = db('prof_trace')
if con:
    rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                            "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                            "from kernels group by name order by sum(end-start) desc"))
    total = sum(r[2] for r in rows)
    with open(os.path.join(DST, tag + '_kernel_trace_stats.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-baseline-seconds 0   (default: 200 timed + 10 warm-up steps)'])
        w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'percent', 'vgpr', 'sgpr', 'lds_bytes',
                    'grid_x', 'workgroup_x'])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), int(r[3]), int(r[4]), int(r[5]), '%.2f' % (100.0 * r[2] / total)] + list(r[6:]))
    print(open(os.path.join(DST, tag + '_kernel_trace_stats.csv')).read())

con = db('prof_allan')
if con:
    rows = list(con.execute("select name, grid_x, grid_y, count(*), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(lds_size) "
                            "from kernels where name like '%allan%' group by name, grid_x, grid_y order by avg(end-start) desc"))
    with open(os.path.join(DST, tag + '_allan_kernel_trace.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['# rocprofv3 --kernel-trace --stats -- python tools/bench_allan.py  (192 series x 1 440 000 samples, 11 calls)'])
        w.writerow(['kernel', 'grid_x', 'grid_y', 'calls', 'avg_ns', 'min_ns', 'max_ns', 'vgpr', 'lds_bytes'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], int(r[4]), int(r[5]), int(r[6]), r[7], r[8]])
    print(open(os.path.join(DST, tag + '_allan_kernel_trace.csv')).read())

pmc = {}
for d in sorted(os.listdir(SRC)):
    if not d.startswith('prof_pmc'):
        continue
    con = db(d)
    if not con:
        continue
    for k, c, n, avg, mn, mx in con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                                           "from counters_collection group by kernel_name, counter_name"):
        pmc.setdefault(k, {})[c] = (n, avg, mn, mx)
with open(os.path.join(DST, tag + '_pmc_counters.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['# rocprofv3 --pmc <one group per pass> --kernel-trace -- python bench.py --steps 10 --warmup 2 '
                '--cpu-baseline-seconds 0 ; values are per dispatch (avg/min/max over dispatches)'])
    w.writerow(['kernel', 'counter', 'dispatches', 'avg', 'min', 'max'])
    for k in sorted(pmc):
        for c in sorted(pmc[k]):
            w.writerow([k, c] + ['%.6g' % v if isinstance(v, float) else v for v in pmc[k][c]])

traffic = {}
for k, c in pmc.items():
    if 'mc_kernel' in k and 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        # MI355X_MICROARCH.md (HBM): rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reads
        # half of the bytes of a coalesced stream -> doubled.  WRITE_SIZE matched the known byte count exactly here.
        fetch_b = 2.0 * c['FETCH_SIZE'][1] * 1024
        write_b = c['WRITE_SIZE'][1] * 1024
        key = 'mc_kernel_rf1_free_given' if re.search(r'mc_kernel<\d, \d, true', k) else 'mc_kernel_rf1_free_keep'
        traffic[key] = {'kernel': k, 'hbm_bytes_per_launch': fetch_b + write_b, 'fetch_bytes_corrected': fetch_b,
                        'write_bytes': write_b, 'source': 'profiles/%s_pmc_counters.csv' % tag}
with open(os.path.join(DST, 'pmc_traffic.json'), 'w') as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(traffic, indent=1))
for k in pmc:
    if 'mc_kernel' in k:
        c = pmc[k]
        if 'SQ_WAVE_CYCLES' in c:
            print(k)
            if 'GRBM_GUI_ACTIVE' in c:
                print('  GRBM_GUI_ACTIVE (summed over the 8 XCDs) per launch: %.4g' % c['GRBM_GUI_ACTIVE'][1])
            print('VALU active / wave cycles: %.3f ; busy cycles %.4g ; VALU insts/wave %.4g' % (
                c['SQ_ACTIVE_INST_VALU'][1] / c['SQ_WAVE_CYCLES'][1], c['SQ_BUSY_CYCLES'][1], c['SQ_INSTS_VALU'][1] / c['SQ_WAVES'][1]))
