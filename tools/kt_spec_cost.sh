#!/bin/bash
# kernel trace of the one-frame bench with the folded accept tests off / on (AVT_SPEC_COST): per-kernel averages, run through gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for V in ${1:-0 1}; do
  rm -rf /tmp/kt_sc_$V
  AVT_SPEC_COST=$V timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_sc_$V -o p -- python $R/bench.py --frames 1 --steps 25 --warmup 2 --regions 3 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 > /tmp/kt_sc_$V.log 2>&1
  echo "== AVT_SPEC_COST=$V"
  python $R/tools/rocpd_stats.py $(find /tmp/kt_sc_$V -name "*.db" | head -1) | cut -c1-140 | sed -n 2,14p
done
