#!/bin/bash
# round 5, call e: C3 statistics in the wave-specialised kernel (A/B at full size), kept-runs overlap, the new tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_process_stats.py tests/test_gpu_c3_long_drive.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
for rep in 1 2; do
  for f in 1 0; do
    GINSIM_SPLIT_PS=$f timeout 300 python tools/experiments/c3_once.py > $OUT/c3_ps${f}_rep$rep.json 2> $OUT/c3_ps${f}_rep$rep.err
    echo "split_ps=$f rep=$rep $(cat $OUT/c3_ps${f}_rep$rep.json)"
  done
done
timeout 300 python tools/experiments/kept_runs_overlap.py > $OUT/kept_overlap.log 2>&1; cat $OUT/kept_overlap.log
