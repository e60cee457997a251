#!/bin/bash
for P in f64 f32; do export AB_PREC=$P; echo "== $P"; for i in 1 2; do for L in gnss-ins-sim_amd/lib/libginsim.so tools/build/libginsim_on.so; do GINSIM_LIB=$PWD/$L python tools/ab_time.py; done; done; done
GINSIM_LIB=$PWD/tools/build/libginsim_on.so python -m pytest tests/test_gpu_edge_cases.py -q -k bitwise 2>&1 | tail -3
