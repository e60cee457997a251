#!/bin/bash
# round 5, call s: the same scan over a 230 GB contiguous arena (where are the fast windows?), twice
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s
mkdir -p $OUT
cd $ROOT
rocm-smi --showmemuse --showmeminfo vram > $OUT/meminfo.txt 2>&1
for i in 1 2; do
  ARENA_GB=230 STEP_GB=2 timeout 300 python tools/experiments/headline_region_scan.py > $OUT/scan_$i.json 2> $OUT/scan_$i.err
  python - <<PY
import json
d = json.load(open('$OUT/scan_$i.json'))
print('arena %s' % d['arena'])
print('  fast offsets: ' + ' '.join('%g' % o for o, m in d['offset_gb__kernel_ms'] if m < 1.28))
print('  ' + ' '.join('%g:%.2f' % (o, m) for o, m in d['offset_gb__kernel_ms']))
PY
  tail -1 $OUT/scan_$i.err | cut -c1-200
done
