#!/bin/bash
# Interleaved A/B of library builds on ONE box, one frame and 64 frames: tools/ab2.sh "<libA> <libB> .." [repeats]   (names relative to avatar_amd/csrc)
# prints every sample and the median per (lib, frames)
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; N=${2:-4}; FRAMES=${3:-"1 64"}
rm -f /tmp/ab2.txt
for rep in $(seq 1 $N); do for F in $FRAMES; do for L in $LIBS; do
  AVT_LIB=$R/avatar_amd/csrc/$L timeout 120 python $R/bench.py --frames $F --steps 20 --warmup 5 --regions 9 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --detail-file /tmp/hl.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', $F, d['ms_per_step'])" >> /tmp/ab2.txt
done; done; done
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for line in open('/tmp/ab2.txt'):
    l, f, v = line.split(); d[(l, int(f))].append(float(v))
for k in sorted(d, key=lambda k: (k[1], k[0])):
    print('%-34s frames %3d  median %.4f  samples %s' % (k[0], k[1], statistics.median(d[k]), ' '.join('%.4f' % x for x in d[k])))
PY
