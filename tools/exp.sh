#!/bin/bash
# scratch experiment driver for one gpurun call (edited per call; results under gpurun_out/exp_<tag>/)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3
timeout 600 python bench.py --cpu-baseline-seconds 0 --pmc off > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
for l in d.get("configs", []):
    if l["name"] == "sim_e2e":
        print(json.dumps(l["C2"])); print(json.dumps(l["C3"]))
PY
