#!/bin/bash
# round 5, call p: the bimodal headline and the way its buffers are allocated: hipMalloc against
# hipExtMallocWithFlags(hipDeviceMallocContiguous) (physically contiguous VRAM -> the largest page-table fragments)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05p
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for fl in 0 4; do
    GINSIM_MALLOC_FLAGS=$fl timeout 300 python tools/experiments/headline_state.py plain plain dummy4 dummy16 plain undummy plain > $OUT/state_f${fl}_$rep.jsonl 2> $OUT/state_f${fl}_$rep.err
    python - <<PY
import json
r = [json.loads(l) for l in open('$OUT/state_f${fl}_$rep.jsonl')]
print('flags $fl rep $rep: ' + ' '.join('%.3f' % d['kernel_ms'] for d in r))
PY
    tail -2 $OUT/state_f${fl}_$rep.err | cut -c1-200
  done
done
