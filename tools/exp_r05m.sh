#!/bin/bash
# round 5, call m: the headline leg alone (bench.py --no-legs) with the library before / after the 11-instruction normal
# transform, INTERLEAVED on one box (the order of two whole bench runs shifted even the legs that draw no normals by 8 %)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05m
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3 4; do
  for lib in libginsim_base.so libginsim.so; do
    GINSIM_LIB=$ROOT/gnss-ins-sim_amd/lib/$lib timeout 300 python bench.py --no-legs --cpu-baseline-seconds 0 --pmc off --no-repeat > $OUT/h_${lib}_$rep.json 2> $OUT/h_${lib}_$rep.err
    python -c "import json;d=json.load(open('$OUT/h_${lib}_$rep.json'));print('%-20s step %.4f ms kernel %.4f ms frac %.3f'%('$lib',d['ms_per_step'],d['roofline']['kernel_ms_avg'],d['roofline']['frac']))"
  done
done
