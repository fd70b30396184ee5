#!/bin/bash
# tools/trace_one_frame.sh [frames] - rocprofv3 kernel trace of the graph replay, per-kernel launch durations (per launch shape).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; F=${1:-1}
rocprofv3 --kernel-trace --stats -d $O/prof_t -o p -- python $R/bench.py --frames $F --steps 25 --warmup 2 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --saturation-frames 0 --regions 3 > $O/prof_t.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_t -name "*.db" | head -1) > $O/trace_${F}_frames.txt
rm -rf $O/prof_t
