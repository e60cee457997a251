#!/bin/bash
# round 5, call b: the fused Allan kernel (levels 0+1 in one launch) -- parity, then A/B against the two-launch form
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_allan.py -x -q > $OUT/allan_tests.log 2>&1; tail -5 $OUT/allan_tests.log
for rep in 1 2 3; do
  for f in 1 0; do
    GINSIM_ALLAN_FUSE=$f timeout 300 python tools/bench_allan.py > $OUT/allan_fuse${f}_rep$rep.json 2>$OUT/allan_fuse${f}_rep$rep.err
    echo "fuse=$f rep=$rep $(python -c "import json;d=json.load(open('$OUT/allan_fuse${f}_rep$rep.json'));print('ms %.4f min %.4f wall %.4f frac %.3f'%(d['ms'],d['ms_min'],d['ms_wall'],d['frac_of_8TBps']))")"
  done
done
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  GINSIM_ALLAN_FUSE=$f timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_fuse$f -o allan -- python $ROOT/tools/bench_allan.py > $OUT/prof_fuse$f.json 2> $OUT/prof_fuse$f.log
  find $OUT/prof_fuse$f -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'echo "--- fuse='$f'"; cut -d, -f1-6 {} | head -8'
done
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_multi_device.py tests/test_leaves_and_fallthrough.py tests/test_gpu_distributed.py -x -q -m gpu > $OUT/new_tests.log 2>&1; tail -5 $OUT/new_tests.log
