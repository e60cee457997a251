#!/bin/bash
export GINSIM_SPLIT=1
for i in 1 2 3; do for L in "$@"; do GINSIM_LIB=$PWD/$L python tools/ab_time.py; done; done
