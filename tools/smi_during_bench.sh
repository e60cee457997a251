(python bench.py --steps 4000 --warmup 10 --cpu-baseline-seconds 0 > /tmp/b.json 2>/dev/null &) 
sleep 4
for i in 1 2 3; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | head -8; echo --; sleep 1.5; done
wait
sleep 6
python - <<PY
import json; d=json.load(open('/tmp/b.json')); print('bench', d['value'], d['roofline']['kernel_ms_avg'])
PY
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
