#!/bin/bash
# per-launch time of k_nn_part with and without its scan (libavatar_hip_nn_noscan.so: no candidate is looked at): the kernel's fixed work
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for fr in ${FRS:-64 512}; do for lib in libavatar_hip.so libavatar_hip_nn_noscan.so; do
  AVT_LIB=$R/avatar_amd/csrc/$lib rocprofv3 --kernel-trace --stats -d $O/prof_fx -o p -- python $R/bench.py --frames $fr ${DENSE:-} --steps 4 --warmup 2 --regions 2 --no-cpu-baseline --no-shard > /dev/null 2>&1
  echo "== frames $fr $lib"; python $R/tools/rocpd_stats.py $(find $O/prof_fx -name "*.db" | head -1) | grep -E "k_nn_part" | cut -c1-140; rm -rf $O/prof_fx
done; done
