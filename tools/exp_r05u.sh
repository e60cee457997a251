#!/bin/bash
# round 5, call u: MonteCarloJob.spread_outputs in the bench: the placement tests, then `bench.py --placement spread` against
# `--placement asis`, interleaved, headline only, three processes each; then the whole line once with the legs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05u
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_placement.py -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed\|Error" $OUT/tests.log | tail -5
for rep in 1 2 3 4; do
  for pl in asis spread; do
    timeout 300 python bench.py --no-legs --cpu-baseline-seconds 0 --pmc off --no-repeat --placement $pl > $OUT/h_${pl}_$rep.json 2> $OUT/h_${pl}_$rep.err
    python - <<PY
import json
d = json.load(open('$OUT/h_${pl}_$rep.json'))
print('%-7s step %.4f ms kernel %.4f ms frac %.3f %s' % ('$pl', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], {k: v for k, v in (d['config'].get('placement') or {}).items() if k != 'note'}))
PY
  done
done


