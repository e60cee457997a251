#!/bin/bash
for S in 0 1; do echo "SPLIT=$S"; GINSIM_SPLIT=$S python tools/store_cost.py; done
