#!/bin/bash
# round 5, call v: pass A without masks in whole steps + a simple-model pass B (series_kernel<3>) against the library before
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05v
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_allan.py tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log | tail -3
VARIANTS="libginsim.so libginsim_base.so" OUTDIR=r05v bash tools/exp_r05k.sh
