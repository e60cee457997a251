#!/bin/bash
# round 5, call q: the headline's planes placed by hand in one contiguous arena: does the time follow the RELATIVE placement of
# the sensor and trajectory planes, or the arena's absolute position?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05q
mkdir -p $OUT
cd $ROOT
i=0
for pre in "" "3" "3,7" "" "50"; do
  i=$((i+1))
  PRE_GB=$pre timeout 300 python tools/experiments/headline_placement.py > $OUT/place_$i.jsonl 2> $OUT/place_$i.err
  python - <<PY
import json
r = [json.loads(l) for l in open('$OUT/place_$i.jsonl')]
print('pre [$pre] arena %s' % r[0].get('arena'))
print('  own %.3f | gaps ' % r[0]['kernel_ms'] + ' '.join('%.3f' % d['kernel_ms'] for d in r if 'gap' in d))
print('  shifted ' + ' '.join('%.3f' % d['kernel_ms'] for d in r if 'shift' in d) + ' | own %.3f' % r[-1]['kernel_ms'])
PY
  tail -1 $OUT/place_$i.err | cut -c1-200
done
