#!/bin/bash
# round 5, call n: the bimodal headline (1.17 / 1.39 ms on one box): per process, re-allocation under several conditions + clocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05n
mkdir -p $OUT
cd $ROOT
rocm-smi --showclocks > $OUT/clocks_before.txt 2>&1
for rep in 1 2 3 4; do
  timeout 300 python tools/experiments/headline_state.py plain plain dummy4 dummy16 dummy64 undummy plain > $OUT/state_$rep.jsonl 2> $OUT/state_$rep.err
  python - <<PY
import json
for l in open('$OUT/state_$rep.jsonl'):
    d = json.loads(l)
    print('rep $rep %-8s %.4f ms (min %.4f) frac %.3f traj %s %s' % (d['condition'], d['kernel_ms'], d['kernel_ms_min'], d['frac'], d['ptrs'].get('traj_free'), d['clocks']))
PY
done
