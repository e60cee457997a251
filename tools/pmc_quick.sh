#!/bin/bash
# usage: tools/pmc_quick.sh <outdir> <counter list in quotes> -- <command...>   : one rocprofv3 --pmc pass, per-kernel averages printed
OUT=$1; CTR=$2; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $CTR --kernel-trace -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3, glob, os
dbs = [os.path.join(r, f) for r, _, fs in os.walk("$OUT") for f in fs if f.endswith('.db')]
con = sqlite3.connect(dbs[0])
rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, n, v in rows:
    if 'mc_kernel' in k or 'allan' in k:
        print('%-60s %-24s n=%d avg=%.4g' % (k.replace('void ginsim::', '').split('(')[0][:60], c, n, v))
PY
