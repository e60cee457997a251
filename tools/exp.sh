#!/bin/bash
# scratch experiment driver for one gpurun call (edited per call; results under gpurun_out/exp_<tag>/)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_allan.py -m gpu -x -q > $OUT/pytest_allan.log 2>&1; tail -2 $OUT/pytest_allan.log
for f in 1 0 1 0; do GINSIM_ALLAN_FUSE=$f timeout 300 python tools/bench_allan.py > $OUT/allan_fuse$f.json 2>&1; tail -1 $OUT/allan_fuse$f.json | cut -c1-130; done
cd /tmp && export TMPDIR=/tmp
for f in 1; do GINSIM_ALLAN_FUSE=$f timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_allan$f -o a -- python $ROOT/tools/bench_allan.py > $OUT/prof_allan$f.log 2>&1; done
cd $ROOT
python - <<PY
import sqlite3, glob
for f in (1,):
    dbs = glob.glob("$OUT/prof_allan%d/**/*.db" % f, recursive=True)
    if not dbs: print('no db', f); continue
    con = sqlite3.connect(dbs[0])
    print('FUSE', f)
    for r in con.execute("select name, grid_x, count(*), avg(end-start), min(end-start) from kernels where name like '%allan%' group by name, grid_x order by avg(end-start) desc"):
        print('   %-60s grid %8d calls %3d avg %.1f us min %.1f us' % (r[0][:60], r[1], r[2], r[3] / 1e3, r[4] / 1e3))
PY
