#!/bin/bash
# quick: bitwise/parity subset + timing of the current build, fp64 and fp32
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp32.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -2
for P in f64 f32; do export AB_PREC=$P; for i in 1 2; do python tools/ab_time.py; done; done
