#!/bin/bash
# round 5, call j: the few-run sensor generation with a lane holding four consecutive samples (one scan per 256 samples) against
# the round-4 form (lane = sample, one scan per 64): series tests, then interleaved A/B timings of the generation
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05j
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
for rep in 1 2 3; do
  for lib in libginsim.so libginsim_base.so; do
    GINSIM_LIB=$ROOT/gnss-ins-sim_amd/lib/$lib timeout 300 python tools/experiments/series_generation.py > $OUT/gen_${lib}_$rep.json 2>$OUT/gen_${lib}_$rep.err
    cat $OUT/gen_${lib}_$rep.json
  done
done
