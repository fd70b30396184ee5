#!/bin/bash
# one counter set on one bench configuration: tools/pmc_one.sh <frames> "<counters>" <kernel name parts...>   (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; F=$1; SET=$2; shift 2
COMMON="--no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --regions 3"
rm -rf $O/prof_one
rocprofv3 --pmc $SET -d $O/prof_one -o p -- python $R/bench.py --frames $F --steps 3 --warmup 1 $COMMON > $O/prof_one.log 2>&1
python $R/tools/pmc_counters.py $(find $O/prof_one -name "*.db" | head -1) "$@"
rm -rf $O/prof_one
