#!/bin/bash
# kernel trace (per launch shape) of one bench configuration: tools/trace_cfg.sh "<bench args>"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_tr -o p -- python $R/bench.py $1 --steps 5 --warmup 2 --regions 3 --no-cpu-baseline --no-shard > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_tr -name "*.db" | head -1) | cut -c1-140 | head -${2:-18}
rm -rf $O/prof_tr
