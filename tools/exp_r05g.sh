#!/bin/bash
# round 5, call g: the wide finishing launch of the Allan call (level 2 + fold + single-chunk levels) -- parity, then A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_allan.py -x -q > $OUT/allan_tests.log 2>&1; tail -5 $OUT/allan_tests.log
for rep in 1 2 3; do
  for f in 1 0; do
    GINSIM_ALLAN_WIDE=$f timeout 300 python tools/bench_allan.py > $OUT/allan_wide${f}_rep$rep.json 2>$OUT/allan_wide${f}_rep$rep.err
    echo "wide=$f rep=$rep $(python -c "import json;d=json.load(open('$OUT/allan_wide${f}_rep$rep.json'));print('ms %.4f min %.4f wall %.4f frac %.3f'%(d['ms'],d['ms_min'],d['ms_wall'],d['frac_of_8TBps']))")"
  done
done
cd /tmp && export TMPDIR=/tmp
GINSIM_ALLAN_WIDE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_wide1 -o allan -- python $ROOT/tools/bench_allan.py > $OUT/prof_wide1.json 2> $OUT/prof_wide1.log
