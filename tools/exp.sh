#!/bin/bash
# scratch experiment driver for one gpurun call (edited per call; results under gpurun_out/exp_<tag>/)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 600 python -X faulthandler -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_allan.py -m gpu -x -q > $OUT/pytest_series.log 2>&1; tail -5 $OUT/pytest_series.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("bench: %.4g %s, %.3f ms/step, kernel %.3f ms, frac %.3f, traffic %s" % (d["value"], d["unit"], d["ms_per_step"],
      d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["roofline"]["traffic"]))
for l in d.get("configs", []):
    if "roofline" in l:
        r = l["roofline"]
        print("  leg %-22s kernel %.3f ms  bound %s frac %s traffic %s" % (l["name"], r["kernel_ms_avg"], r["bound"], r["frac"], r.get("traffic")))
    if l["name"] == "C5_allan_end_to_end":
        print("     gen %.3f ms (min %.3f)  relayout+allan wall %.3f ms  allan min %.3f  layout %s  T/A %s" % (l["sensor_generation_ms"], l["sensor_generation_ms_min"], l["relayout_plus_allan_wall_ms"], l["allan_call_ms_min"], l["sensor_layout"], l["roofline"].get("traffic_over_algorithmic")))
    if l["name"] == "sim_e2e":
        print("     ", json.dumps(l)[:600])
PY
