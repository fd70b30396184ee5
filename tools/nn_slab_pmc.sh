#!/bin/bash
# SQ counters of k_nn_part with and without the slab scan
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for slab in 1 0; do
  extra="AVT_X=1"; [ $slab = 0 ] && extra="AVT_NN_NO_SLAB=1"
  echo "== slab $slab"
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS"; do
    env $extra rocprofv3 --pmc $set -d $O/prof_np -o p -- python $R/bench.py --frames ${FR:-64} --steps 2 --warmup 1 --regions 1 --no-cpu-baseline --no-shard > /dev/null 2>&1
    python $R/tools/pmc_counters.py $(find $O/prof_np -name "*.db" | head -1) k_nn_part | grep "192x32\|224x256" | cut -c1-120
    rm -rf $O/prof_np
  done
done
