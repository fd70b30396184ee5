#!/bin/bash
# eval workgroups per frame (AVT_G) against frames per GPU: ms per step of bench.py.  FRS="16 64 128 512" TS="384 512 640 768 1024 1536"
for fr in ${FRS:-16 24 64 128 512}; do
  nfg=$(python -c "print($fr if $fr < 32 else ($fr + 1) // 2)")
  for T in ${TS:-384 512 640 768 1024 1536}; do
    g=$(( T / nfg )); [ $g -lt 2 ] && continue; [ $g -gt 128 ] && continue
    AVT_G=$g python bench.py --frames $fr --steps 6 --warmup 2 --regions 5 --no-cpu-baseline --no-shard 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $fr target $T G=$g', d['ms_per_step'], d['value'], d['roofline']['launch_shape']['eval_workgroups_per_frame'])"
  done
done
