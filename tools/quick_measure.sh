#!/bin/bash
# tools/quick_measure.sh [frames ...] - bench.py without the side legs: it/s, ms per step and the per-class kernel times
# of the instrumented pass for each frame count (default 1 and 64).  Run on the GPU box (gpurun -- 'tools/quick_measure.sh').
for F in ${@:-1 64}; do
python bench.py --frames $F --steps 30 --warmup 2 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage \
    --no-shard --saturation-frames 0 --regions 7 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('F=$F', round(d['value'], 1), 'it/s', round(d['ms_per_step'], 4), 'ms |', ' '.join('%s %.4f' % (n, v['ms']) for n, v in k.items()))"
done
