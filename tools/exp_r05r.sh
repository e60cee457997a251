#!/bin/bash
# round 5, call r: launch time of the headline against the position of its planes in a 96 GB physically contiguous arena
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05r
mkdir -p $OUT
cd $ROOT
i=0
for pre in "" "" "40"; do
  i=$((i+1))
  PRE_GB=$pre timeout 300 python tools/experiments/headline_region_scan.py > $OUT/scan_$i.json 2> $OUT/scan_$i.err
  python - <<PY
import json
d = json.load(open('$OUT/scan_$i.json'))
print('pre [%s] arena %s' % (d['pre_gb'], d['arena']))
print('  ' + ' '.join('%g:%.2f' % (o, m) for o, m in d['offset_gb__kernel_ms']))
PY
  tail -1 $OUT/scan_$i.err | cut -c1-200
done
STEP_GB=0.25 ARENA_GB=24 timeout 300 python tools/experiments/headline_region_scan.py > $OUT/scan_fine.json 2> $OUT/scan_fine.err
python - <<PY
import json
d = json.load(open('$OUT/scan_fine.json'))
print('fine: arena %s' % d['arena'])
print('  ' + ' '.join('%g:%.2f' % (o, m) for o, m in d['offset_gb__kernel_ms']))
PY
