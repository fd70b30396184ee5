#!/bin/bash
# A/B of two builds of the library in one gpurun call: tools/lib_ab.sh <libA> <libB> "<bench args>" [repeats]
for i in $(seq 1 ${4:-3}); do for lib in $1 $2; do
  AVT_LIB=$PWD/$lib python bench.py $3 --steps 8 --warmup 2 --regions 7 --no-cpu-baseline --no-shard 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$3', d['ms_per_step'], d['value'])"
done; done
