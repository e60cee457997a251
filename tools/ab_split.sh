#!/bin/bash
for i in 1 2 3; do for S in 0 1; do echo -n "SPLIT=$S "; GINSIM_SPLIT=$S python tools/ab_time.py; done; done
