#!/bin/bash
# end-to-end ms per step with and without the slab scan (AVT_NN_NO_SLAB=1) for the batch configurations
for cfg in "64" "128" "512" "16 --dense" "64 --dense"; do for off in 0 1; do
  extra=""; [ $off = 1 ] && extra="AVT_NN_NO_SLAB=1"
  env $extra python bench.py --frames $cfg --steps 6 --warmup 2 --regions 5 --no-cpu-baseline --no-shard 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $cfg slab', 'off' if $off else 'on ', d['ms_per_step'], d['value'], d['kernels']['nn']['ms'])"
done; done
