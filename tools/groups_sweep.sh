#!/bin/bash
# frame groups per optimize() call (AVT_GROUPS) at a few batch sizes
for F in ${@:-32 64 128 512}; do for g in 1 2 3 4; do echo -n "groups=$g "; AVT_GROUPS=$g bash tools/quick_measure.sh $F 2>&1 | tail -1 | cut -c1-60; done; done
