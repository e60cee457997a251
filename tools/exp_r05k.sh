#!/bin/bash
# round 5, call k: variants of the few-run sensor generation (samples per lane x wavefronts per SIMD): HIP-event time of the
# three launches and the per-kernel durations from a kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${OUTDIR:-r05k}
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in ${VARIANTS:-libginsim.so libginsim_base.so}; do
    GINSIM_LIB=$ROOT/gnss-ins-sim_amd/lib/$lib timeout 300 python tools/experiments/series_generation.py > $OUT/gen_${lib}_$rep.json 2>$OUT/gen_${lib}_$rep.err
    python -c "import json;d=json.load(open('$OUT/gen_${lib}_$rep.json'));print('%-22s %.4f ms (min %.4f)'%(d['lib'],d['generation_ms'],d['generation_ms_min']))"
  done
done
for lib in ${VARIANTS:-libginsim.so libginsim_base.so}; do
  GINSIM_LIB=$ROOT/gnss-ins-sim_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace -d $OUT/trace_$lib -o t -- python tools/experiments/series_generation.py > $OUT/trace_$lib.log 2>&1
  python tools/experiments/trace_summary.py $OUT/trace_$lib '%series_%' | tee $OUT/trace_$lib.json
  rm -rf $OUT/trace_$lib
done
