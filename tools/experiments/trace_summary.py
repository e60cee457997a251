"""Per-kernel average duration from a rocprofv3 --kernel-trace output directory (its sqlite database): one JSON line."""
import glob
import json
import os
import sqlite3
import sys

d = sys.argv[1]
like = sys.argv[2] if len(sys.argv) > 2 else '%'
hits = sorted(glob.glob(os.path.join(d, '**', '*.db'), recursive=True))
con = sqlite3.connect(hits[0])
rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(vgpr_count) from kernels where name like ? "
                   "group by name order by sum(end-start) desc", (like,)).fetchall()
print(json.dumps({'dir': os.path.basename(d.rstrip('/')),
                  'kernels': [{'name': r[0][:60], 'calls': r[1], 'avg_us': r[2] / 1e3, 'min_us': r[3] / 1e3, 'vgpr': r[4]} for r in rows]}))
