#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ab
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python tools/experiments/sim_c3.py > $OUT/sim_c3.log 2>&1
grep "^rep" $OUT/sim_c3.log
python - <<PY
import glob, sqlite3
db = sorted(glob.glob('$OUT/trace/**/*.db', recursive=True))[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels where end - start > 2000000 order by start").fetchall()
t0 = rows[0][1]
for r in rows:
    print('%-60s start %9.3f ms  end %9.3f ms  dur %8.3f ms grid %d wg %d' % (r[0][:60], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], r[4]))
PY
rm -rf $OUT/trace
