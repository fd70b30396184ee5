#!/bin/bash
# start / duration / gap of the kernels of the LAST optimize() of a bench configuration (both frame groups interleaved)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace -d $O/prof_tl -o p -- python $R/bench.py $1 --steps 3 --warmup 2 --regions 1 --no-cpu-baseline --no-shard > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find $O/prof_tl -name "*.db" | head -1) ${2:-100}
rm -rf $O/prof_tl
