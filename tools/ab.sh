#!/bin/bash
# usage: tools/ab.sh libA.so libB.so   (paths relative to the repo root) -- alternates the two builds 3 times
for i in 1 2 3; do
  for L in "$@"; do GINSIM_LIB=$PWD/$L python tools/ab_time.py; done
done
