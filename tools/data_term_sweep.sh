#!/bin/bash
# rows against moments over the frames-per-GPU axis (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
COMMON="--steps 5 --warmup 2 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --regions 5"
for F in ${FRAMES:-16 32 64 96 128 192 256 384 512}; do
  for dt in rows moments; do
    python $R/bench.py --frames $F --data-term $dt $COMMON 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('frames %4d  %-8s %8.4f ms/step  %10.0f GN it/s' % ($F, '$dt', d['ms_per_step'], d['value']))"
  done
done
