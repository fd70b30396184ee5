#!/bin/bash
# same-box A/B of two builds of the library on the one-frame bench (kernel trace): tools/kt_lib_ab.sh "<lib a> <lib b> .." [env assignments]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; shift
i=0
for L in $LIBS; do      # every library twice, alternating: run-to-run spread beside the difference
  i=$((i+1)); rm -rf /tmp/kt_ab_$i
  env "$@" AVT_LIB=$R/avatar_amd/csrc/$L timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_ab_$i -o p -- python $R/bench.py --frames 1 --steps 25 --warmup 2 --regions 3 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 > /tmp/kt_ab_$i.log 2>&1
  echo "== $L  ($(grep -o '"ms_per_step":[0-9.]*' /tmp/kt_ab_$i.log | head -1))"
  python $R/tools/rocpd_stats.py $(find /tmp/kt_ab_$i -name "*.db" | head -1) | cut -c1-128 | grep -E "k_solveILi256ELb0ELi[12]|k_evalILi24ELi10ELi6ELb0|k_nn_vis|k_records"
done
