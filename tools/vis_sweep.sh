#!/bin/bash
# visibility launch shape of frame batches: per-face workgroups (AVT_VIS_FRAME_MIN=0) against one workgroup per frame (=1)
for m in 0 1; do for F in ${@:-16 64 512}; do echo -n "min=$m "; AVT_VIS_FRAME_MIN=$m bash tools/quick_measure.sh $F 2>&1 | tail -1 | cut -c1-150; done; done
