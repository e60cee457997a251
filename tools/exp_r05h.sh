#!/bin/bash
# round 5, call h: does spinning on hipStreamQuery shorten the Allan call (host wake-up latency behind the last kernel)?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for f in 1 0; do
    GINSIM_SPIN_WAIT=$f timeout 300 python tools/bench_allan.py > $OUT/allan_spin${f}_rep$rep.json 2>$OUT/allan_spin${f}_rep$rep.err
    echo "spin=$f rep=$rep $(python -c "import json;d=json.load(open('$OUT/allan_spin${f}_rep$rep.json'));print('ms %.4f min %.4f wall %.4f frac %.3f'%(d['ms'],d['ms_min'],d['ms_wall'],d['frac_of_8TBps']))")"
  done
done
timeout 600 python -m pytest tests/test_gpu_allan.py -x -q > $OUT/allan_tests.log 2>&1; tail -2 $OUT/allan_tests.log
