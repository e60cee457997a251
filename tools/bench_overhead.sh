#!/bin/bash
# per-step overhead of bench.py beyond the MC kernel: N = 1, and N = 2 ranks sharing GPU 0 over gloo (control-flow check)
for i in 1 2; do python bench.py --cpu-baseline-seconds 0 --steps 40 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('N=1 ms_per_step %.4f kernel %.4f overhead %.1f us value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], 1e3*(d['ms_per_step']-d['roofline']['kernel_ms_avg']), d['value']))"; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --backend gloo --shared-device --runs-per-gpu 32768 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('N=2 shared ms_per_step %.4f kernel %.4f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value']))"
