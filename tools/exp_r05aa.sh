#!/bin/bash
# round 5, call aa: the kept runs of a statistics-only Sim as the first workgroup of the batch (_BlockAndRest): the Sim tests, then
# the end-to-end leg of the bench (C3 as named: 9-axis + GPS, two kept runs)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05aa
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_process_stats.py tests/test_gpu_sim_dropin.py tests/test_gpu_full_size.py tests/test_gpu_multi_device.py tests/test_gpu_c3_long_drive.py tests/test_gpu_plugin_surface.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed\|Error" $OUT/tests.log | tail -5
timeout 600 python - > $OUT/sim_e2e.json 2> $OUT/sim_e2e.err <<PY
import json, sys, os
sys.path.insert(0, os.path.join('$ROOT', 'gnss-ins-sim_amd')); sys.path.insert(0, '$ROOT')
import bench
from ginsim import workloads
print(json.dumps(bench.leg_sim_e2e(workloads)))
PY
python - <<PY
import json
l = json.load(open('$OUT/sim_e2e.json'))
for t in ('C2', 'C3'):
    print('sim %s run %.4f s results %.4f s  %.4g sample*MC/s  walls %s' % (t, l[t]['run_wall_s'], l[t]['results_wall_s'], l[t]['sample_MC_per_s_end_to_end'], ['%.3f' % w for w in l[t]['wall_s_every_construction']]))
PY
tail -3 $OUT/sim_e2e.err
