#!/bin/bash
# k_compact / k_nn_part with and without the slab scan (AVT_NN_NO_SLAB): kernel-trace means per launch shape
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for fr in ${FRS:-64 512}; do for slab in 1 0; do
  extra=""; [ $slab = 0 ] && extra="AVT_NN_NO_SLAB=1"
  env $extra rocprofv3 --kernel-trace --stats -d $O/prof_ab_${fr}_$slab -o p -- python $R/bench.py --frames $fr ${DENSE:-} --steps 4 --warmup 2 --regions 2 --no-cpu-baseline --no-shard > /dev/null 2>&1
  echo "== frames $fr slab $slab"
  python $R/tools/rocpd_stats.py $(find $O/prof_ab_${fr}_$slab -name "*.db" | head -1) | grep -E "k_compact|k_nn_part" | cut -c1-140
  rm -rf $O/prof_ab_${fr}_$slab
done; done
