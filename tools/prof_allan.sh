#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_allan -o allan -- python $ROOT/tools/bench_allan.py > $OUT/prof_allan.log 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob('$OUT/prof_allan/*.db'):
    con = sqlite3.connect(db)
    for r in con.execute("select name, count(*), avg(end-start), min(end-start), max(grid_x), max(grid_y) from kernels group by name, grid_x order by avg(end-start) desc limit 12"):
        print(r)
PY
