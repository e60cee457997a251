#!/bin/bash
# round 5, call l: two samples per lane in the few-run generation + the 11-instruction normal transform (LDS table in 256-byte
# octave blocks, v_bfi sign) against the library before both (libginsim_base.so): all GPU tests, the generation A/B with a
# kernel trace, and the whole bench with either library on the same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05l
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log | tail -3
VARIANTS="libginsim.so libginsim_base.so libginsim_s2w4.so" OUTDIR=r05l bash tools/exp_r05k.sh
for lib in libginsim.so libginsim_base.so; do
  GINSIM_LIB=$ROOT/gnss-ins-sim_amd/lib/$lib timeout 600 python bench.py --cpu-baseline-seconds 0 --pmc off --no-repeat > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err
  python tools/show_bench.py $OUT/bench_$lib.json | head -16
done
