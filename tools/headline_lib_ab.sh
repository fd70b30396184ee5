#!/bin/bash
# the bench's headline step (one frame, hipGraph replay, median of 15 regions of 50 steps) for several builds of the library on ONE box, alternating:
# tools/headline_lib_ab.sh "<lib> .." [env assignments]
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; shift
for rep in 1 2; do for L in $LIBS; do
  env "$@" AVT_LIB=$R/avatar_amd/csrc/$L timeout 120 python $R/bench.py --frames 1 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --detail-file /tmp/hl.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s ms_per_step %.4f  value %.0f' % ('$L', d['ms_per_step'], d['value']))"
done; done
