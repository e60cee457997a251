#!/bin/bash
for S in 0 1 2; do echo "SPLIT=$S"; GINSIM_SPLIT=$S python tools/variant_checksum.py; done
for i in 1 2; do for S in 0 1 2; do echo -n "SPLIT=$S "; GINSIM_SPLIT=$S python tools/ab_time.py; done; done
