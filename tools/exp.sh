#!/bin/bash
# scratch experiment driver for one gpurun call (edited per call; results under gpurun_out/exp_<tag>/)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 1200 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 300 python tools/sim_profile.py > $OUT/sim_profile.log 2>&1; head -70 $OUT/sim_profile.log
