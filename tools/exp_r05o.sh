#!/bin/bash
# round 5, call o: is the slow state of the headline a DRAM bank conflict between the 15 output planes (plane stride 500 MiB
# exactly: equal low address bits in every plane)?  The same launch with 65 472 runs (row 511.5 KB: planes skewed by 4 KB steps)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05o
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for runs in 65536 65472; do
    RUNS=$runs timeout 300 python tools/experiments/headline_state.py plain plain dummy4 dummy16 plain > $OUT/state_${runs}_$rep.jsonl 2> $OUT/state_${runs}_$rep.err
    python - <<PY
import json
for l in open('$OUT/state_${runs}_$rep.jsonl'):
    d = json.loads(l)
    print('runs $runs rep $rep %-8s %.4f ms (min %.4f) frac %.3f' % (d['condition'], d['kernel_ms'], d['kernel_ms_min'], d['frac']))
PY
  done
done
